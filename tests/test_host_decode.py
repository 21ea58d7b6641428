"""CPU checks of the decode code that k_backend runs on the GPU.

backend_kernels.h writes the decoder state machine and its numpy restatements
(histogram "auto", percentile, pairwise mean) as __host__ __device__ functions;
csrc/selftest.h exports host-compiled hooks to them so this file can compare
the very same source against numpy and against oracle/decoders.py without a
GPU.  The GPU parity tests (test_gpu_*.py) repeat the comparisons through the
real kernels.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import decoders as dec
from sdr_amd import _lib, synth


@pytest.fixture(scope="module")
def L():
    return _lib.load()


def _mid_edge(L, x):
    x = np.ascontiguousarray(x, np.float64)
    out = C.c_double()
    assert L.sdr_selftest_hist_mid_edge(x.ctypes.data, x.size, C.byref(out)) == 0
    return out.value


def test_hist_auto_mid_edge_matches_numpy(L):
    rng = np.random.default_rng(1)
    cases = [np.array([]), np.array([5.0]), np.array([5.0, 5.0, 5.0]), np.array([1.0, 2.0]),
             np.array([400.0] * 30 + [900.0] * 34), np.array([400.0] * 40 + [900.0] * 24),
             np.arange(64, dtype=np.float64), np.arange(67, dtype=np.float64) * 50.0]
    for n in list(range(1, 40)) + [64, 67, 100, 128, 255, 300, 1000]:
        k = n // 2
        x = np.concatenate([400 + 30 * rng.standard_normal(k), 900 + 30 * rng.standard_normal(n - k)])
        cases.append(np.round(x / 50.0) * 50.0)   # quantised like 20 kHz sampling
        cases.append(x)
    for x in cases:
        e = np.histogram(x, "auto")[1]
        assert _mid_edge(L, x) == e[e.size // 2], x


def test_hist_auto_with_zero_iqr_is_numpy_2(L):
    """np.histogram(x, "auto") when the middle half of the widths is one value (IQR == 0; plausible at 20 kHz,
    where widths are multiples of 50 us and a packet can be mostly one symbol).  The oracle and the library
    reproduce numpy >= 1.15 (here 2.2.6): the Freedman-Diaconis width is 0, so the Sturges width alone sets the
    bins.  The reference is Python 2 and ran an older numpy, where a zero "auto" width meant ONE bin and
    edges[size // 2] was the maximum -- a different pulse_short_limit.  Nothing in the reference pins either
    (its README packet has IQR > 0): this case is parity unpinned, and the test states which behaviour is built."""
    cases = [np.array([400.0] * 50 + [900.0] * 14), np.array([450.0] * 60 + [950.0] * 7),
             np.array([400.0] * 81 + [900.0] * 3 + [950.0] * 2), np.array([500.0] * 9 + [1000.0])]
    for x in cases:
        q75, q25 = np.percentile(x, [75, 25])
        assert q75 - q25 == 0.0
        e = np.histogram(x, "auto")[1]
        assert _mid_edge(L, x) == e[e.size // 2]
        assert e.size > 2 and x.min() < e[e.size // 2] < x.max()      # not the single-bin edge of numpy < 1.15


def test_ts33c_packet_of_mostly_one_symbol(L):
    """The decoder-level form of the case above: 81 short pulses and 9 long ones (two widths among the long
    ones).  Library (host-compiled device code) == oracle, limit between the classes; under the one-bin rule
    the limit would be the longest pulse and the 900 us pulses would flip to 1."""
    sym = synth.ts33c_symbols([1] * 81 + [0] * 9, None)
    k = [i for i, (lvl, w) in enumerate(sym) if lvl == 1 and w > 700.0]
    sym[k[0]] = (1, sym[k[0]][1] + 50.0)                            # one long pulse a sample longer
    env = np.concatenate([synth.envelope(sym, 20000, lead_in_us=3000.0), np.zeros(100, np.uint8)]).astype(np.int8)
    o = dec.TS33CStreamDecoder(sample_rate=20000)
    o.work([env])
    g = HostDec(L, _lib.DEC_TS33C, 20000)
    g.work(env)
    assert len(o.frames) == len(g.frames) == 1
    assert_ts33c_equal(o.frames[0], g.frames[0])
    lim = o.frames[0]["pulse_short_limit"]
    assert 450.0 < lim < 900.0 and o.frames[0]["nbits"] == 90
    assert o.frames[0]["packet"] == bytes(9) + b"\xff"
    g.close()


def test_pow_tables_match_numpy(L):
    """n**(-1/3) and log2(n) feed the bin width; the library tabulates them with
    libm on the host -- must equal what numpy computes on this host."""
    for n in list(range(1, 300)) + [512, 1000, 1024, 2047]:
        x = np.linspace(0.0, 1.0, n)
        e = np.histogram(x, "auto")[1]
        assert _mid_edge(L, x) == e[e.size // 2]


def test_pairwise_mean_matches_numpy(L):
    rng = np.random.default_rng(2)
    for n in list(range(1, 128)):
        x = np.ascontiguousarray(rng.random(n) * 60.0)
        out = C.c_double()
        assert L.sdr_selftest_pairwise_mean(x.ctypes.data, n, C.byref(out)) == 0
        assert out.value == np.mean(x), n


class HostDec(object):
    def __init__(self, L, decoder, rate):
        self.L = L
        self.p = L.sdr_selftest_dec_create(decoder, float(rate))
        self.frames = []
        self.state = None

    def work(self, chunk):
        chunk = np.ascontiguousarray(chunk, np.int8)
        buf = (_lib.Frame * 64)()
        n = C.c_size_t()
        st = (C.c_int32 * 4)()
        rc = self.L.sdr_selftest_dec_work(self.p, chunk.ctypes.data, chunk.size, buf, 64, C.byref(n), C.byref(st))
        assert rc == 0
        self.frames += [_copy(buf[i]) for i in range(n.value)]
        self.state = list(st)

    def close(self):
        self.L.sdr_selftest_dec_destroy(self.p)


def _copy(f):
    g = _lib.Frame()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(g))
    return g


def assert_ts33c_equal(of, gf):
    assert _lib.STATUS_TEXT[gf.status] == of["status"]
    assert gf.nbits == of["nbits"]
    assert gf.lim[0] == of["pulse_short_limit"] and gf.lim[1] == of["gap_short_limit"]
    if of["packet"] is not None:
        assert gf.packet() == of["packet"][:16]
    if of["status"] == "ok":
        assert gf.v[0] == of["channel"] and gf.v[1] == of["rolling_code"]
        assert bool(gf.v[2]) == of["battery_ok"] and gf.v[3] == of["humidity"]
        assert gf.temp_tenths / 10. == of["temperature"]


def assert_tfa_equal(of, gf):
    assert _lib.STATUS_TEXT[gf.status] == of["status"]
    assert gf.lim[0] == of["symbol_length"] or (np.isnan(gf.lim[0]) and np.isnan(of["symbol_length"]))
    if of["packet"] is not None:
        assert gf.packet() == of["packet"]
    if of["status"] == "ok":
        assert "{:02X}".format(gf.v[0]) == of["identifier"]
        assert gf.v[1] == of["humidity"] and bool(gf.v[2]) == of["battery_ok"]
        p = gf.packet()
        assert (p[4] & 0x0f) * 10 + ((p[5] >> 4) & 0x0f) * 1 + (p[5] & 0x0f) * 0.1 - 40 == of["temperature"]


def _ts33c_stream(rng, n_frames=4, rate=20000, noise=True):
    sym = []
    pkts = []
    for _ in range(n_frames):
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        pkts.append(pkt)
        sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng if noise else None)
        if rng.random() < 0.3:
            sym += [(0, float(rng.integers(2000, 60000)))]
    lead = float(rng.integers(0, 8000))
    env = synth.envelope(sym, rate, lead_in_us=lead)
    env = np.concatenate([env, np.zeros(int(rng.integers(0, 400)), np.uint8)])
    return env.astype(np.int8), pkts, lead


def _random_chunks(rng, x, mode):
    if mode == "one":
        return [x]
    if mode == "split100":
        return np.array_split(x, 100)
    cuts = np.sort(rng.integers(0, x.size + 1, int(rng.integers(1, 60))))
    return np.split(x, cuts)


@pytest.mark.parametrize("mode", ["one", "split100", "random"])
@pytest.mark.parametrize("seed", range(6))
def test_ts33c_host_decode_vs_oracle(L, seed, mode):
    rng = np.random.default_rng(100 + seed)
    bits, pkts, lead = _ts33c_stream(rng)
    if seed == 5:  # corrupt: flip a stretch so parity / length errors occur
        bits = bits.copy()
        k = int(rng.integers(200, bits.size - 200))
        bits[k:k + 12] ^= 1
    o = dec.TS33CStreamDecoder(sample_rate=20000)
    g = HostDec(L, _lib.DEC_TS33C, 20000)
    for ch in _random_chunks(rng, bits, mode):
        o.work([ch])
        g.work(ch)
        assert g.state == [int(o.state == "frame"), int(o.currentSymbols.size), int(o.previousSymbolType), o.chunk_id]
    assert len(g.frames) == len(o.frames)
    for of, gf in zip(o.frames, g.frames):
        assert_ts33c_equal(of, gf)
    if mode == "one" and seed != 5:
        got = [f["packet"] for f in o.frames if f["status"] == "ok"]
        # a capture that starts less than reset_limit before the first pulse
        # loses that frame in the reference too (TS33C/stream_decoder.py:104-105)
        assert got == (pkts if lead > 1600 else pkts[len(pkts) - len(got):]) and len(got) >= len(pkts) - 1
    g.close()


@pytest.mark.parametrize("mode", ["one", "split200", "random"])
@pytest.mark.parametrize("seed", range(4))
def test_tfa_host_decode_vs_oracle(L, seed, mode):
    rng = np.random.default_rng(300 + seed)
    sym = []
    pkts = []
    for _ in range(3):
        pkt = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10.0,
                               humidity=int(rng.integers(0, 100)), battery_ok=bool(rng.integers(0, 2)))
        pkts.append(pkt)
        sym += synth.tfa_symbols(synth.tfa_bits(pkt), rng, reset_gap_us=4000.0)
    env = synth.envelope(sym, 2e6, lead_in_us=900.0)
    env = np.concatenate([env, np.zeros(1500, np.uint8)]).astype(np.int8)
    o = dec.TFAStreamDecoder(sample_rate=2000000)
    g = HostDec(L, _lib.DEC_TFA, 2000000)
    chunks = np.array_split(env, 200) if mode == "split200" else _random_chunks(rng, env, mode)
    for ch in chunks:
        o.work([ch])
        g.work(ch)
        assert g.state == [int(o.state == "frame"), int(o.currentSymbols.size), int(o.previousSymbolType), o.chunk_id]
    assert len(g.frames) == len(o.frames) == 3
    for of, gf in zip(o.frames, g.frames):
        assert_tfa_equal(of, gf)
    assert [f["packet"] for f in o.frames] == pkts
    g.close()


def test_tfa_error_paths(L):
    """odd sync count -> the reference raises ValueError (reshape); no data start
    -> IndexError; empty sync -> NaN guard (GRC twin) / ValueError (script)."""
    rate = 2000000
    def run(decoder, widths_us):
        sym = [(1 - (i & 1), w) for i, w in enumerate(widths_us)]
        env = synth.envelope(sym, rate, lead_in_us=600.0).astype(np.int8)
        g = HostDec(L, decoder, rate)
        g.work(env)
        fr = g.frames
        g.close()
        return env, fr
    # 6 junk widths, then 2 widths + reset gap = 3 (odd) -> reshape fails
    env, fr = run(_lib.DEC_TFA, [20, 20, 20, 20, 20, 20, 15, 11, 700])
    assert [f.status for f in fr] == [_lib.FRAME_EXC_RESHAPE]
    o = dec.TFAStreamDecoder(sample_rate=rate)
    with pytest.raises(ValueError):
        o.work([env])
    # 2 widths after the junk: mean pair sum 715 us, nothing exceeds 1.5x -> IndexError
    env, fr = run(_lib.DEC_TFA, [20, 20, 20, 20, 20, 20, 15, 700])
    assert [f.status for f in fr] == [_lib.FRAME_EXC_INDEX]
    o = dec.TFAStreamDecoder(sample_rate=rate)
    with pytest.raises(IndexError):
        o.work([env])
    # 4 widths after the junk: decodes to too few bits
    env, fr = run(_lib.DEC_TFA, [20, 20, 20, 20, 20, 20, 15, 11, 15, 700])
    assert [f.status for f in fr] == [_lib.FRAME_BAD_LENGTH]
    o = dec.TFAStreamDecoder(sample_rate=rate)
    o.work([env])
    assert [f["status"] for f in o.frames] == ["Invalid packet length"]
    # even sync but nothing longer than 1.5 symbol before the reset gap... the gap itself is
    env, fr = run(_lib.DEC_TFA, [20, 20, 20, 20, 20, 20, 700])
    assert [f.status for f in fr] == [_lib.FRAME_EXC_RESHAPE]
    # exactly 6 widths then the reset gap: sync = [gap] -> odd
    env, fr = run(_lib.DEC_TFA_GRC, [20, 20, 20, 20, 20, 700])
    assert [f.status for f in fr] == [_lib.FRAME_SYMBOL_LENGTH]
    o = dec.TFAStreamDecoder(sample_rate=rate, nan_guard=True)
    o.work([env])
    assert [f["status"] for f in o.frames] == ["Symbol length error"]
    env, fr = run(_lib.DEC_TFA, [20, 20, 20, 20, 20, 700])
    assert [f.status for f in fr] == [_lib.FRAME_EXC_NAN]
    o = dec.TFAStreamDecoder(sample_rate=rate)
    with pytest.raises(ValueError):
        o.work([env])


def _events_decode(cls, env, rate, wd):
    s = cls()
    for lvl, tick in dec.edges_to_events(env, rate, watchdog_us=wd):
        s.cbf(lvl, tick)
    return s.frames


@pytest.mark.parametrize("seed", range(4))
def test_rule_decoders_host_vs_oracle(L, seed, kat):
    rng = np.random.default_rng(500 + seed)
    rate = 1000000
    # somfy
    fr = synth.somfy_frame(0xA0, int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 24)))
    env = synth.envelope(synth.somfy_symbols(fr, repeats=int(rng.integers(0, 3))), rate, lead_in_us=3000.0)
    env = _jitter_env(rng, env)
    of = _events_decode(dec.SomfySniffer, env, rate, 20000)
    g = HostDec(L, _lib.DEC_SOMFY, rate)
    g.work(env.astype(np.int8))
    assert len(g.frames) == len(of) >= 1
    for a, b in zip(of, g.frames):
        assert bytes(bytearray(b.raw[:7])) == a["raw"] and b.packet() == a["frame"]
        assert (b.status == 0) == a["checksum_ok"]
        assert (b.v[0], b.v[1], b.v[2], b.v[3]) == (a["control"], a["rolling_code"], a["address"], a["clock"])
    g.close()
    # berner
    code = int(rng.integers(0, 1 << 32))
    env = _jitter_env(rng, synth.envelope(synth.berner_symbols(code, repeats=5), rate, lead_in_us=3000.0))
    of = _events_decode(dec.BernerSniffer, env, rate, None)
    g = HostDec(L, _lib.DEC_BERNER, rate)
    g.work(env.astype(np.int8))
    assert [f.packet() for f in g.frames] == of and len(of) == 3
    g.close()
    # mumbi
    name = sorted(kat["mumbi_codes"])[seed]
    env = _jitter_env(rng, synth.envelope(synth.mumbi_symbols(bytes(kat["mumbi_codes"][name]), repeats=4), rate, lead_in_us=9000.0))
    of = _events_decode(dec.MumbiSniffer, env, rate, 8000)
    g = HostDec(L, _lib.DEC_MUMBI, rate)
    g.work(env.astype(np.int8))
    assert [f.packet() for f in g.frames] == of and len(of) == 4
    g.close()
    # kwmobile (Nexus PPM)
    bits = synth.kwmobile_bits(int(rng.integers(0, 256)), int(rng.integers(0, 3)), bool(rng.integers(0, 2)),
                               int(rng.integers(0, 600)), int(rng.integers(0, 100)))
    env = _jitter_env(rng, synth.envelope(synth.kwmobile_symbols(bits, repeats=5), rate, lead_in_us=4000.0))
    of = _events_decode(dec.KwmobileDecoder, env, rate, 3000)
    g = HostDec(L, _lib.DEC_KWMOBILE, rate)
    g.work(env.astype(np.int8))
    assert len(of) == 4 and [f.packet() for f in g.frames] == [f["frame"] for f in of]
    for a, b in zip(of, g.frames):
        assert (b.v[0], b.v[1], bool(b.v[2]), b.v[3]) == (a["sensor_id"], a["humidity"], a["battery_ok"], a["channel"])
        assert b.temp_tenths / 10. == a["temperature"] and b.nbits == 36
    g.close()
    # baldr
    bits = synth.baldr_bits(int(rng.integers(0, 256)), int(rng.integers(0, 4)), int(rng.integers(0, 2)),
                            int(rng.integers(-400, 600)), int(rng.integers(0, 100)))
    env = _jitter_env(rng, synth.envelope(synth.baldr_symbols(bits), rate, lead_in_us=3000.0))
    of = _events_decode(dec.BaldrDecoder, env, rate, 18000)
    g = HostDec(L, _lib.DEC_BALDR, rate)
    g.work(env.astype(np.int8))
    assert len(of) == 1 and [f.packet() for f in g.frames] == [f["frame"] for f in of]
    for a, b in zip(of, g.frames):
        assert (b.v[0], b.v[1], b.v[2], b.v[3]) == (a["sensor_id"], a["humidity"], a["txMode"], a["channel"])
        assert b.temp_tenths / 10. == a["temperature"] and b.nbits == 37
    g.close()


def _jitter_env(rng, env):
    """move every edge by -2..+2 samples (keeps the run order)."""
    pos = np.flatnonzero(np.diff(env.astype(np.int8))) + 1
    out = np.zeros_like(env)
    lvl = int(env[0])
    start = 0
    for p in pos:
        q = int(p + rng.integers(-2, 3))
        q = max(start + 1, min(q, env.size))
        out[start:q] = lvl
        lvl ^= 1
        start = q
    out[start:] = lvl
    return out


def test_velux_host_vs_oracle(L, kat):
    """rule_velux (host-compiled) against the oracle restatement: the reference's two captures,
    truncations (the script's unpack errors), bit flips (CRC) and random prefixes."""
    rng = np.random.default_rng(77)
    base = kat["velux_active"]["bits"]
    cases = [base, kat["velux_commented"]["bits"], base[:150], base[:380], base[:372], base[:371], base[:130]]
    for _ in range(6):
        b = list(rng.integers(0, 2, int(rng.integers(0, 40)))) + list(base[80:])
        if rng.random() < 0.5:
            b[int(rng.integers(130, len(b)))] ^= 1
        cases.append(b)
    for bits in cases:
        o = dec.velux_decode(bits)
        g = HostDec(L, _lib.DEC_VELUX, 1)
        g.work(np.asarray(bits, np.int8))
        if o is None:
            assert g.frames == []
        else:
            assert len(g.frames) == 1
            f = g.frames[0]
            assert int(f.lim[0]) == o["frame_start"]
            if o["error"] == "short":
                assert f.status == _lib.FRAME_EXC_SHORT
            else:
                assert f.packet() == o["packet"]
                assert (f.status == _lib.FRAME_OK) == (o["crc_ok"] and o["error"] is None)
                if o["error"] is None:
                    assert f.temp_tenths == o["counter"] and int(f.lim[1]) == o["message_length"]
        g.close()


def _rule_stream(rng, kind, rate):
    """several transmissions of one emitter with idle gaps and junk between them"""
    sym = []
    for _ in range(int(rng.integers(2, 4))):
        if kind == "somfy":
            fr = synth.somfy_frame(0xA0 | int(rng.integers(0, 16)), int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 65536)),
                                   int(rng.integers(0, 1 << 24)))
            sym += synth.somfy_symbols(fr, repeats=int(rng.integers(0, 3)), wakeup=bool(rng.integers(0, 2)))
        elif kind == "berner":
            sym += synth.berner_symbols(int(rng.integers(0, 1 << 32)), repeats=int(rng.integers(1, 5)))
        elif kind == "mumbi":
            codes = sorted(synth.MUMBI_CODES)
            sym += synth.mumbi_symbols(bytes(bytearray(synth.MUMBI_CODES[codes[int(rng.integers(0, len(codes)))]])),
                                       repeats=int(rng.integers(1, 5)), on_air=bool(rng.integers(0, 2)))
        elif kind == "kwmobile":
            sym += synth.kwmobile_symbols(synth.kwmobile_bits(int(rng.integers(0, 256)), int(rng.integers(0, 3)), True,
                                                              int(rng.integers(-300, 500)), int(rng.integers(0, 100))),
                                          repeats=int(rng.integers(2, 6)))
        else:
            sym += synth.baldr_symbols(synth.baldr_bits(int(rng.integers(0, 256)), int(rng.integers(0, 3)), int(rng.integers(0, 2)),
                                                        int(rng.integers(-300, 500)), int(rng.integers(0, 100))),
                                       repeats=int(rng.integers(1, 4)))
        r = rng.random()
        if r < 0.4:
            for _ in range(int(rng.integers(1, 20))):
                sym += [(1, float(rng.integers(30, 1200))), (0, float(rng.integers(30, 1200)))]
        sym += [(0, float(rng.integers(500, 45000)))]
    env = synth.envelope(sym, rate, lead_in_us=float(rng.integers(0, 9000))).astype(np.int8)
    env = np.concatenate([env, np.zeros(int(rng.integers(0, 30000)), np.int8)])
    if rng.random() < 0.25:
        env = (1 - env).astype(np.int8)
    return env


RULE_SET = [("somfy", _lib.DEC_SOMFY, dec.SomfySniffer, 20000), ("berner", _lib.DEC_BERNER, dec.BernerSniffer, None),
            ("mumbi", _lib.DEC_MUMBI, dec.MumbiSniffer, 8000), ("kwmobile", _lib.DEC_KWMOBILE, dec.KwmobileDecoder, 3000),
            ("baldr", _lib.DEC_BALDR, dec.BaldrDecoder, 18000)]


def rule_frame_key(decoder, f):
    """comparable form of a device/host frame record"""
    if decoder == _lib.DEC_SOMFY:
        return (bytes(bytearray(f.raw[:7])), f.packet(), f.status == 0, f.v[3])
    return f.packet()


def oracle_frame_key(decoder, a):
    if decoder == _lib.DEC_SOMFY:
        return (a["raw"], a["frame"], a["checksum_ok"], a["clock"])
    if decoder in (_lib.DEC_KWMOBILE, _lib.DEC_BALDR):
        return a["frame"]
    return a


@pytest.mark.parametrize("seed", range(6))
def test_rule_decoders_carry_state_across_calls(L, seed):
    """The pigpio callbacks are continuous (somfy/sniff.py:28-117, berner/sniff.py:21-44, mumbi/sniff.py:26-65,
    kwmobile/decoder.py:95-135, baldr/decoder.py:87-142): a frame cut by a call boundary must still decode.
    Every emitter stream runs through EVERY rule in ragged chunks (cuts inside pulses, inside frames, empty
    chunks, one-byte chunks); the frames of all calls together equal the oracle sniffer fed the whole stream."""
    rng = np.random.default_rng(900 + seed)
    total = 0
    for kind, _, _, _ in RULE_SET:
        rate = int(rng.choice([100000, 250000, 1000000]))
        env = _rule_stream(rng, kind, rate)
        for _, decoder, cls, wd in RULE_SET:
            want = [oracle_frame_key(decoder, a) for a in _events_decode(cls, env, rate, wd)]
            for mode in ("ragged", "tiny", "one"):
                if mode == "one":
                    cuts = [0, env.size]
                elif mode == "tiny":
                    cuts = sorted(set([0, env.size] + [int(v) for v in rng.integers(0, env.size, 60)]))
                else:
                    cuts = sorted(set([0, env.size] + [int(v) for v in rng.integers(0, env.size, 7)]))
                    cuts = cuts[:3] + [cuts[2]] + cuts[3:]          # an empty call
                    if env.size > 10:
                        cuts = sorted(cuts + [cuts[1] + 1])          # a one-byte call
                g = HostDec(L, decoder, rate)
                for a, b in zip(cuts[:-1], cuts[1:]):
                    g.work(np.ascontiguousarray(env[a:b]))
                got = [rule_frame_key(decoder, f) for f in g.frames]   # HostDec keeps the frames of all calls
                g.close()
                assert got == want, (kind, decoder, mode, rate)
            total += len(want)
    assert total >= 5
