"""GPU parity tests: the HIP path through the C ABI (sdr_amd.chain.Chain ->
libsdrhip.so) against the CPU oracle on the same seeded inputs.

Bar: filter/demod floats bit-identical to the oracle (stated tolerance 1e-5
relative, observed 0 -- same fmaf order); sliced bytes, frames and decoder
state bit-exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import _lib, chain, synth
from tests.test_host_decode import assert_tfa_equal, assert_ts33c_equal

pytestmark = pytest.mark.gpu

FS = 1e6
REL_TOL = 1e-5


def _taps(fs=FS):
    return chain.low_pass(1.0, fs, 20e3, 40e3, chain.WIN_HAMMING)


def _assert_float_parity(got, want):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    scale = np.max(np.abs(want)) if want.size else 1.0
    err = np.max(np.abs(got - want)) if want.size else 0.0
    assert err <= REL_TOL * scale, (err, scale)
    return int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))) if got.dtype == np.float32 else 0


def test_firdes_matches_oracle():
    for args in [(1, 2e6, 20e3, 40e3, 0), (1, 2e6, 100e3, 50e3, 0), (1, 2e6, 400e3, 200e3, 2),
                 (1, 2e6, 500e3, 250e3, 2), (2.5, 2e6, 500e3, 1e6, 2), (1, 5e5, 23437.5, 1562.5, 0)]:
        a = chain.low_pass(*args)
        b = odsp.low_pass(*args)
        assert a.size == b.size and np.array_equal(a, b)
    assert [chain.low_pass(1, 2e6, 20e3, 40e3).size, chain.low_pass(1, 2e6, 100e3, 50e3).size] == [121, 97]
    for l, m in [(1, 4), (1, 10), (48, 50)]:
        assert np.array_equal(chain.resampler_taps(l, m), odsp.resampler_taps(l, m))
    assert [chain.resampler_taps(1, 4).size, chain.resampler_taps(1, 10).size, chain.resampler_taps(48, 50).size] == [131, 329, 821]


@pytest.mark.parametrize("D,ntaps_fs", [(1, 2e6), (4, 2e6), (50, 1e6), (64, 1e6), (1000, 1e6)])
def test_fir_ccf_float_parity(D, ntaps_fs):
    rng = np.random.default_rng(7 + D)
    n = 20000 + 13
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    taps = _taps(ntaps_fs)
    with chain.Chain(FS, decim=D, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_samples=n) as c:
        c.batch_process(x[None, :])
        y = c.fetch_demod()
    want = odsp.fir_ccf(x, taps, D)
    assert y.size == want.size == n // D
    bad = _assert_float_parity(y.view(np.float32), want.view(np.float32))
    assert bad == 0


@pytest.mark.parametrize("D", [1, 50])
def test_xlating_mag2_float_and_bits_parity(D):
    rng = np.random.default_rng(11)
    n = 1 << 17
    iq, pkt, env = synth.ts33c_capture(n, FS, seed=33, f_off=100e3)
    taps = _taps()
    keep = 50 // D
    with chain.Chain(FS, decim=D, center_freq=100e3, taps=taps, demod=chain.DEMOD_MAG2, demod_gain=1.0,
                     thr=(0.05, 0.15), keep_n=keep, tap_demod=True, max_samples=n) as c:
        c.batch_process(iq[None, :])
        mag = c.fetch_demod()
        bits = c.fetch_bits()
    wbits, wmag = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, keep, fc=100e3, fs=FS, return_mag=True)
    assert _assert_float_parity(mag, wmag) == 0
    assert np.array_equal(bits, wbits)
    assert bits.sum() > 100
    del rng


def test_nco_vs_gnuradio_rotator_short():
    """nco64 (exact phase) vs the GNU Radio float recurrence on a short stream:
    within 1e-5 relative; the recurrence itself drifts beyond that on long ones."""
    rng = np.random.default_rng(3)
    n = 4096
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    with chain.Chain(2e6, decim=1, center_freq=-200e3, demod=chain.DEMOD_NONE, tap_demod=True, max_samples=n) as c:
        c.batch_process(x[None, :])
        y = c.fetch_demod()
    want = odsp.freq_xlating_fir(x, [1.0], -200e3, 2e6, 1, rotator="nco64")
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))
    gr = odsp.freq_xlating_fir(x, [1.0], -200e3, 2e6, 1, rotator="gr")
    assert np.max(np.abs(y - gr)) <= 1e-5 * np.max(np.abs(gr))


def _oracle_ts33c(bits):
    d = odec.TS33CStreamDecoder(sample_rate=20000)
    d.work([bits])
    return d


@pytest.mark.parametrize("shape", ["F", "R"])
def test_ts33c_batch_frames_bit_exact(shape):
    ncap, n = 6, 1 << 18
    D, keep = (50, 1) if shape == "F" else (1, 50)
    taps = _taps()
    rng = np.random.default_rng(5)
    iqs, pkts = [], []
    for i in range(ncap):
        iq, pkt, _ = synth.ts33c_capture(n, FS, seed=1000 + i, **synth.random_ts33c_fields(rng))
        iqs.append(iq)
        pkts.append(pkt)
    iqs = np.stack(iqs)
    with chain.Chain(FS, decim=D, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, max_batch=ncap, max_samples=n) as c:
        out = c.batch_process(iqs)
        bits = c.fetch_bits().reshape(ncap, -1)
        frames = c.fetch_frames()
    assert out.n_bits == n // 50
    k = 0
    for i in range(ncap):
        wbits = odsp.ook_chain(iqs[i], taps, D, 1.0, 0.05, 0.15, keep)
        assert np.array_equal(bits[i], wbits), i
        o = _oracle_ts33c(wbits)
        mine = [f for f in frames if f.capture == i]
        assert len(mine) == len(o.frames) == 2  # the third frame is cut off by the capture end
        for of, gf in zip(o.frames, mine):
            assert_ts33c_equal(of, gf)
            assert gf.packet() == pkts[i]
        k += len(mine)
    assert k == len(frames)


def test_streaming_iq_equals_one_shot():
    """FIR history, NCO phase, hysteresis bit, keep_one_in_n counter and decoder
    state carried across sdr_process_iq calls: ragged chunks == one call."""
    n = 1 << 18
    iq, pkt, _ = synth.ts33c_capture(n, FS, seed=77, f_off=50e3)
    taps = _taps()
    rng = np.random.default_rng(9)
    for D, keep in [(50, 1), (5, 10)]:
        wbits = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, keep, fc=50e3, fs=FS)
        cuts = np.sort(rng.integers(0, n, 23))
        chunks = np.split(iq, cuts) + [np.zeros(0, np.complex64)]
        o = odec.TS33CStreamDecoder(sample_rate=20000)
        with chain.Chain(FS, decim=D, center_freq=50e3, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C,
                         max_samples=n) as c:
            got, frames = [], []
            for ch in chunks:
                c.process_iq(ch)
                b = c.fetch_bits()
                got.append(b)
                frames += [_copy(f) for f in c.fetch_frames()]
                o.work([b])
                assert c.decoder_state() == [int(o.state == "frame"), int(o.currentSymbols.size),
                                             int(o.previousSymbolType), o.chunk_id]
        got = np.concatenate(got)
        assert np.array_equal(got, wbits[:got.size]) and wbits.size - got.size <= 1
        assert len(frames) == len(o.frames) == 2
        for of, gf in zip(o.frames, frames):
            assert_ts33c_equal(of, gf)
            assert gf.packet() == pkt


def _copy(f):
    g = _lib.Frame()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(g))
    return g


def test_process_bits_is_streamdecoder_work(kat):
    """sdr_process_bits == StreamDecoder.work([chunk]) incl. the README vector."""
    k = kat["ts33c_readme"]
    runs = [np.zeros(600, np.int8)]
    for p, g in zip(k["pulses"], k["gaps"]):
        runs += [np.ones(p, np.int8), np.zeros(g, np.int8)]
    bits = np.concatenate(runs * 2 + [np.zeros(3000, np.int8)])
    o = odec.TS33CStreamDecoder(sample_rate=250000)
    with chain.Chain(250000, decoder=chain.DEC_TS33C, decoder_rate=250000, max_samples=bits.size) as c:
        frames = []
        for ch in np.array_split(bits, 100) + [np.zeros(0, np.int8)]:
            c.process_bits(ch)
            o.work([ch])
            frames += [_copy(f) for f in c.fetch_frames()]
            assert c.decoder_state() == [int(o.state == "frame"), int(o.currentSymbols.size),
                                         int(o.previousSymbolType), o.chunk_id]
    assert len(frames) == len(o.frames) == 2
    for of, gf in zip(o.frames, frames):
        assert_ts33c_equal(of, gf)
        assert gf.packet().hex() == k["packet_hex"] and gf.temp_tenths == 216 and gf.v[3] == 39


def test_quad_demod_parity():
    """TFA/decoder.grc discriminator branch: LPF(27 taps) -> quadrature_demod."""
    rng = np.random.default_rng(21)
    n = 50000
    ph = np.cumsum(rng.standard_normal(n) * 0.3)
    x = (np.exp(1j * ph) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    taps = chain.low_pass(1, 2e6, 500e3, 250e3, chain.WIN_BLACKMAN)
    gain = 2e6 / (2 * np.pi * 300000 / 8)
    for D in (1, 3):
        with chain.Chain(2e6, decim=D, taps=taps, demod=chain.DEMOD_QUAD, demod_gain=gain, tap_demod=True,
                         thr=(-0.1, 0.1), max_samples=n) as c:
            c.batch_process(x[None, :])
            y = c.fetch_demod()
            bits = c.fetch_bits()
        f = odsp.fir_ccf(x, taps, D)
        want = odsp.quad_demod(f, np.float32(gain))
        assert _assert_float_parity(y, want) == 0
        wb, _, _ = odsp.slice_bits(want, 1.0, -0.1, 0.1, 1)
        assert np.array_equal(bits, wb)


def test_tfa_analyzer2_chain():
    """TFA/analyzer2.grc:3151-3232: xlating(+500k, taps 1) -> mag^2 -> LPF fff(7)
    -> threshold .2/.4 -> slicer -> keep 1 -> decoder @2 MS/s."""
    fs = 2e6
    rng = np.random.default_rng(30)
    pkts, sym = [], []
    for _ in range(2):
        p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10,
                             humidity=int(rng.integers(0, 100)))
        pkts.append(p)
        sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=900.0)
    env = synth.envelope(sym, fs, lead_in_us=800.0)
    env = np.concatenate([env, np.zeros(2500, np.uint8)])
    iq = synth.ook_iq(env, fs, f_off=500e3, amp=0.9, sigma=0.01, rng=rng)
    post = chain.low_pass(1.0, fs, 500e3, 1e6, chain.WIN_BLACKMAN)
    assert post.size == 7
    with chain.Chain(fs, decim=1, center_freq=500e3, demod=chain.DEMOD_MAG2, demod_gain=1.0, thr=(0.2, 0.4),
                     keep_n=1, post_taps=post, tap_demod=True, decoder=chain.DEC_TFA_GRC, max_samples=iq.size) as c:
        c.batch_process(iq[None, :])
        v = c.fetch_demod()
        bits = c.fetch_bits()
        frames = c.fetch_frames()
    y = odsp.freq_xlating_fir(iq, [1.0], 500e3, fs, 1)
    wv = odsp.fir_fff(odsp.mag2(y), post, 1)
    assert _assert_float_parity(v, wv) == 0
    wb, _, _ = odsp.slice_bits(wv, 1.0, 0.2, 0.4, 1)
    assert np.array_equal(bits, wb)
    o = odec.TFAStreamDecoder(sample_rate=2000000, nan_guard=True)
    o.work([wb])
    assert len(frames) == len(o.frames) == 2
    for of, gf, p in zip(o.frames, frames, pkts):
        assert_tfa_equal(of, gf)
        assert gf.packet() == p


def test_tfa_analyzer2_chain_streaming():
    """The same flowgraph fed in ragged work() chunks: the fff filter's history, the
    threshold bit and the decoder's rawBuffer carry over; floats, bits and frames
    equal the one-shot oracle."""
    fs = 2e6
    rng = np.random.default_rng(31)
    pkts, sym = [], []
    for _ in range(2):
        p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10,
                             humidity=int(rng.integers(0, 100)))
        pkts.append(p)
        sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=4000.0)
    env = synth.envelope(sym, fs, lead_in_us=800.0)
    env = np.concatenate([env, np.zeros(9000, np.uint8)])
    iq = synth.ook_iq(env, fs, f_off=500e3, amp=0.9, sigma=0.01, rng=rng)
    post = chain.low_pass(1.0, fs, 500e3, 1e6, chain.WIN_BLACKMAN)
    y = odsp.freq_xlating_fir(iq, [1.0], 500e3, fs, 1)
    wv = odsp.fir_fff(odsp.mag2(y), post, 1)
    wb, _, _ = odsp.slice_bits(wv, 1.0, 0.2, 0.4, 1)
    cuts = [0, 3, 4, 5, 2000, 2001, 9000, iq.size // 2, iq.size // 2 + 2, iq.size]
    o = odec.TFAStreamDecoder(sample_rate=2000000, nan_guard=True)
    got_v, got_b, frames = [], [], []
    with chain.Chain(fs, decim=1, center_freq=500e3, demod=chain.DEMOD_MAG2, demod_gain=1.0, thr=(0.2, 0.4),
                     keep_n=1, post_taps=post, tap_demod=True, decoder=chain.DEC_TFA_GRC, max_samples=iq.size) as c:
        for a, b in zip(cuts[:-1], cuts[1:]):
            c.process_iq(iq[a:b])
            got_v.append(c.fetch_demod())
            got_b.append(c.fetch_bits())
            frames += c.fetch_frames()
            o.work([wb[a:b]])
            st = c.decoder_state()
            assert st[0] == (1 if o.state == "frame" else 0) and st[3] == o.chunk_id
    assert _assert_float_parity(np.concatenate(got_v), wv) == 0
    assert np.array_equal(np.concatenate(got_b), wb)
    assert len(frames) == len(o.frames) == 2
    for of, gf, p in zip(o.frames, frames, pkts):
        assert_tfa_equal(of, gf)
        assert gf.packet() == p


def test_rule_decoders_from_iq(kat):
    """configs[3] emitters: somfy / berner / mumbi envelopes -> IQ -> chain -> frames."""
    fs = 1e6
    taps = _taps()
    rng = np.random.default_rng(40)
    cases = []
    fr = synth.somfy_frame(0xA0, 2, 0x1234, 0xABCDEF)
    cases.append((chain.DEC_SOMFY, synth.somfy_symbols(fr, repeats=1), 3000.0, odec.SomfySniffer, 20000))
    cases.append((chain.DEC_BERNER, synth.berner_symbols(0x02EAABBF), 3000.0, odec.BernerSniffer, None))
    cases.append((chain.DEC_MUMBI, synth.mumbi_symbols(bytes(kat["mumbi_codes"]["c_on"]), repeats=4), 9000.0,
                  odec.MumbiSniffer, 8000))
    kw = [int(ch) for ch in kat["kwmobile_sniff"]["bits"]]
    cases.append((chain.DEC_KWMOBILE, synth.kwmobile_symbols(kw, repeats=4), 4000.0, odec.KwmobileDecoder, 3000))
    cases.append((chain.DEC_BALDR, synth.baldr_symbols(synth.baldr_bits(0x3C, 1, 0, 217, 48)), 3000.0,
                  odec.BaldrDecoder, 18000))
    for decoder, sym, lead, cls, wd in cases:
        env = synth.envelope(sym, fs, lead_in_us=lead)
        iq = synth.ook_iq(env, fs, rng=rng)
        D = 10
        with chain.Chain(fs, decim=D, taps=taps, decoder=decoder, max_samples=iq.size, max_frames=32) as c:
            c.batch_process(iq[None, :])
            bits = c.fetch_bits()
            frames = c.fetch_frames()
        wb = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, 1)
        assert np.array_equal(bits, wb)
        s = cls()
        for lvl, tick in odec.edges_to_events(wb, int(fs / D), watchdog_us=wd):
            s.cbf(lvl, tick)
        assert len(frames) == len(s.frames) >= 1
        for a, b in zip(s.frames, frames):
            if decoder == chain.DEC_SOMFY:
                assert b.packet() == a["frame"] and bytes(bytearray(b.raw[:7])) == a["raw"] == fr
                assert b.status == 0 and b.v[1] == 0x1234 and b.v[2] == 0xABCDEF
            elif decoder in (chain.DEC_KWMOBILE, chain.DEC_BALDR):
                assert b.packet() == a["frame"] and b.temp_tenths / 10. == a["temperature"]
                assert (b.v[0], b.v[1], b.v[3]) == (a["sensor_id"], a["humidity"], a["channel"])
            else:
                assert b.packet() == a
        if decoder == chain.DEC_KWMOBILE:
            assert [f.packet().hex() for f in frames] == [kat["kwmobile_sniff"]["frame_hex"]] * 3
        if decoder == chain.DEC_BALDR:
            assert len(frames) == 1 and frames[0].temp_tenths == 217 and frames[0].v[1] == 48


def test_edge_cases():
    taps = _taps()
    with chain.Chain(FS, decim=50, taps=taps, decoder=chain.DEC_TS33C, max_batch=2, max_samples=4096) as c:
        # all-zero capture: no frames, bits all zero
        c.batch_process(np.zeros((2, 4096), np.complex64))
        assert c.fetch_frames() == [] and not c.fetch_bits().any()
        # too large -> capacity error, handle stays usable
        with pytest.raises(_lib.SdrError) as e:
            c.batch_process(np.zeros((3, 4096), np.complex64))
        assert e.value.code == _lib.SDR_E_CAPACITY
        # ragged: fewer samples than D -> zero outputs
        out = c.batch_process(np.zeros((1, 30), np.complex64))
        assert out.n_bits == 0 and c.fetch_bits().size == 0
    # saturated capture (carrier always on): one run, hysteresis carry crosses every tile
    n = 1 << 16
    iq = np.full(n, 0.7 + 0j, np.complex64)
    with chain.Chain(FS, decim=1, taps=taps, keep_n=7, max_samples=n) as c:
        c.batch_process(iq[None, :])
        b = c.fetch_bits()
    wb = odsp.ook_chain(iq, taps, 1, 1.0, 0.05, 0.15, 7)
    assert np.array_equal(b, wb) and b[-1] == 1


def test_tfa_error_paths_on_gpu():
    """The reference's latent exceptions come back as frame status codes."""
    rate = 2000000
    cases = [(_lib.DEC_TFA, [20, 20, 20, 20, 20, 20, 15, 11, 700], _lib.FRAME_EXC_RESHAPE),
             (_lib.DEC_TFA, [20, 20, 20, 20, 20, 20, 15, 700], _lib.FRAME_EXC_INDEX),
             (_lib.DEC_TFA, [20, 20, 20, 20, 20, 20, 15, 11, 15, 700], _lib.FRAME_BAD_LENGTH),
             (_lib.DEC_TFA_GRC, [20, 20, 20, 20, 20, 700], _lib.FRAME_SYMBOL_LENGTH),
             (_lib.DEC_TFA, [20, 20, 20, 20, 20, 700], _lib.FRAME_EXC_NAN)]
    for decoder, widths, want in cases:
        sym = [(1 - (i & 1), w) for i, w in enumerate(widths)]
        env = synth.envelope(sym, rate, lead_in_us=600.0).astype(np.int8)
        with chain.Chain(rate, decoder=decoder, decoder_rate=rate, max_samples=env.size) as c:
            c.process_bits(env)
            assert [f.status for f in c.fetch_frames()] == [want]


def test_tfa_stream_chunked_on_gpu():
    rate = 2000000
    rng = np.random.default_rng(31)
    pkts, sym = [], []
    for _ in range(3):
        p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10,
                             humidity=int(rng.integers(0, 100)), battery_ok=bool(rng.integers(0, 2)))
        pkts.append(p)
        sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=4000.0)
    env = np.concatenate([synth.envelope(sym, rate, lead_in_us=900.0), np.zeros(1500, np.uint8)]).astype(np.int8)
    o = odec.TFAStreamDecoder(sample_rate=rate)
    frames = []
    with chain.Chain(rate, decoder=chain.DEC_TFA, decoder_rate=rate, max_samples=env.size) as c:
        for ch in np.array_split(env, 200):
            c.process_bits(ch)
            o.work([ch])
            frames += [_copy(f) for f in c.fetch_frames()]
            assert c.decoder_state() == [int(o.state == "frame"), int(o.currentSymbols.size),
                                         int(o.previousSymbolType), o.chunk_id]
    assert len(frames) == len(o.frames) == 3
    for of, gf, p in zip(o.frames, frames, pkts):
        assert_tfa_equal(of, gf)
        assert gf.packet() == p


def test_mumbi_fs300_two_stage_chain(kat):
    """mumbi/FS300.grc:3157-3242: LPF(100k, 97 taps) -> rational_resampler 1/10 (329 taps)
    -> mag^2 -> threshold .1/.4 -> slicer -> keep_one_in_n(20).  Two cascaded channel
    filters (sdr_set_prefilter); floats and bits equal the block-by-block oracle, batch
    and streaming; with keep 2 (100 kHz) the mumbi rule decodes the code."""
    fs = 2e6
    rng = np.random.default_rng(77)
    code = bytes(kat["mumbi_codes"]["c_on"])
    sym = synth.mumbi_symbols(code, repeats=3)
    env = synth.envelope(sym, fs, lead_in_us=9000.0)
    iq = synth.ook_iq(env, fs, amp=0.9, sigma=0.02, rng=rng)
    lpf = chain.low_pass(1.0, fs, 100e3, 50e3)
    rs = chain.resampler_taps(1, 10)
    assert (lpf.size, rs.size) == (97, 329)
    y1 = odsp.fir_ccf(iq, lpf, 1)
    y2 = odsp.resample(y1, rs, 1, 10)
    assert np.array_equal(y2.view(np.uint32), odsp.fir_ccf(y1, rs, 10).view(np.uint32)[:2 * y2.size])
    wv = odsp.mag2(y2)
    for keep_n in (20, 2):
        wb, _, _ = odsp.slice_bits(wv, 1.0, 0.1, 0.4, keep_n)
        with chain.Chain(fs, pre_taps=lpf, pre_decim=1, decim=10, taps=rs, demod=chain.DEMOD_MAG2, thr=(0.1, 0.4),
                         keep_n=keep_n, tap_demod=True, decoder=chain.DEC_MUMBI, max_samples=iq.size) as c:
            c.batch_process(iq[None, :])
            v, bits, frames = c.fetch_demod(), c.fetch_bits(), c.fetch_frames()
            assert _assert_float_parity(v, wv[:v.size]) == 0 and v.size == iq.size // 10
            assert np.array_equal(bits, wb[:bits.size])
            if keep_n == 2:
                o = odec.MumbiSniffer()
                for lvl, tick in odec.edges_to_events(wb, fs / 10 / keep_n, 8000):
                    o.cbf(lvl, tick)
                assert [f.packet() for f in frames] == o.frames and code in o.frames
        # streaming, ragged chunks (both stages keep their own history)
        with chain.Chain(fs, pre_taps=lpf, pre_decim=1, decim=10, taps=rs, demod=chain.DEMOD_MAG2, thr=(0.1, 0.4),
                         keep_n=keep_n, tap_demod=True, max_samples=iq.size) as c:
            cuts = [0, 7, 8, 5000, 5003, 40000, iq.size]
            gv, gb = [], []
            for a, b in zip(cuts[:-1], cuts[1:]):
                c.process_iq(iq[a:b])
                gv.append(c.fetch_demod())
                gb.append(c.fetch_bits())
            assert _assert_float_parity(np.concatenate(gv), wv[:iq.size // 10]) == 0
            assert np.array_equal(np.concatenate(gb), wb[:sum(x.size for x in gb)])


def test_tfa_fsk_discriminator_chain():
    """BASELINE.json configs[2] shape: 2-level FSK at 38400 Bd -> LPF (27 taps, TFA/decoder.grc:1163-1234)
    -> quadrature_demod (TFA/decoder.grc:450-481) -> threshold -> slicer -> TFA stream decoder
    (symbol length from the sync pairs, CRC-8).  The carrier idles on the "0" frequency so the
    discriminator output has the gaps the decoder's reset_limit needs (the reference's own FSK tail
    is disabled in the GRC; its working TFA decode is the mag^2 graph tested above)."""
    fs, dev = 2e6, 60e3
    rng = np.random.default_rng(38)
    pkts, sym = [], []
    for _ in range(2):
        p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10,
                             humidity=int(rng.integers(0, 100)))
        pkts.append(p)
        sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=900.0)
    env = synth.envelope(sym, fs, lead_in_us=800.0)
    env = np.concatenate([env, np.zeros(2500, np.uint8)]).astype(np.float64)
    ph = 2 * np.pi * np.cumsum(dev * (2 * env - 1)) / fs
    iq = (0.8 * np.exp(1j * ph) + 0.01 * (rng.standard_normal(env.size) + 1j * rng.standard_normal(env.size)))
    iq = iq.astype(np.complex64)
    taps = chain.low_pass(1, fs, 500e3, 250e3, chain.WIN_BLACKMAN)
    gain = np.float32(fs / (2 * np.pi * dev))
    with chain.Chain(fs, decim=1, taps=taps, demod=chain.DEMOD_QUAD, demod_gain=gain, thr=(-0.2, 0.2),
                     tap_demod=True, decoder=chain.DEC_TFA_GRC, max_samples=iq.size) as c:
        c.batch_process(iq[None, :])
        v = c.fetch_demod()
        bits = c.fetch_bits()
        frames = c.fetch_frames()
    want = odsp.quad_demod(odsp.fir_ccf(iq, taps, 1), gain)
    assert _assert_float_parity(v, want) == 0
    wb, _, _ = odsp.slice_bits(want, 1.0, -0.2, 0.2, 1)
    assert np.array_equal(bits, wb)
    # the start-up transient of the discriminator is a junk "packet": the GRC twin of the
    # decoder reports it as "Symbol length error" (TFA/analyzer2.grc:959-960)
    o = odec.TFAStreamDecoder(sample_rate=2000000, nan_guard=True)
    o.work([wb])
    assert len(frames) == len(o.frames)
    for of, gf in zip(o.frames, frames):
        if of["status"] == "Symbol length error":
            assert gf.status == _lib.FRAME_SYMBOL_LENGTH
        else:
            assert_tfa_equal(of, gf)
    assert [f.packet() for f in frames if f.status == 0] == pkts


@pytest.mark.parametrize("extra", [0, 1, 15, 16383, 16384, 70001])
def test_long_row_segment_parallel_run_length(extra, monkeypatch):
    """Rows of 128 KiB and more: k_backend resolves the carries, k_runs_seg finds the run starts of
    every segment with its own workgroup, k_backend gathers and decodes.  Same frames as the
    single-workgroup pass (SDR_NO_SEG=1) and as the oracle decoder, also when the row ends inside a
    round or exactly on a segment boundary, and through the whole chain (carry patch first)."""
    rate = 2000000
    rng = np.random.default_rng(77)
    pkts, sym = [], []
    for _ in range(14):
        p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10,
                             humidity=int(rng.integers(0, 100)), battery_ok=bool(rng.integers(0, 2)))
        pkts.append(p)
        sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=4000.0)
    env = synth.envelope(sym, rate, lead_in_us=900.0)
    n = 4 * 65536 + extra
    assert env.size + 1500 < n
    env = np.concatenate([env, np.zeros(n - env.size, np.uint8)]).astype(np.int8)
    o = odec.TFAStreamDecoder(sample_rate=rate)
    o.work([env])
    got = {}
    for mode in ("seg", "single"):
        if mode == "single":
            monkeypatch.setenv("SDR_NO_SEG", "1")
        with chain.Chain(rate, decoder=chain.DEC_TFA, decoder_rate=rate, max_samples=env.size, max_frames=32) as c:
            c.process_bits(env)
            got[mode] = [_copy(f) for f in c.fetch_frames()]
            state = c.decoder_state()
        assert state == [int(o.state == "frame"), int(o.currentSymbols.size), int(o.previousSymbolType), o.chunk_id]
        assert len(got[mode]) == len(o.frames) == 14
        for of, gf, p in zip(o.frames, got[mode], pkts):
            assert_tfa_equal(of, gf)
            assert gf.packet() == p
    monkeypatch.delenv("SDR_NO_SEG")
    # the same row through the whole OOK chain: the hysteresis carries are patched by the first launch
    iq = synth.ook_iq(env.astype(np.uint8), rate, f_off=0.0, amp=0.9, sigma=0.01, rng=rng)
    taps = chain.low_pass(1.0, rate, 500e3, 1e6, chain.WIN_BLACKMAN)
    with chain.Chain(rate, decim=1, taps=taps, thr=(0.2, 0.4), decoder=chain.DEC_TFA_GRC, max_samples=iq.size,
                     max_frames=32) as c:
        c.batch_process(iq[None, :])
        bits = c.fetch_bits()
        frames = [_copy(f) for f in c.fetch_frames()]
    wb = odsp.ook_chain(iq, taps, 1, 1.0, 0.2, 0.4, 1)
    assert np.array_equal(bits, wb)
    o2 = odec.TFAStreamDecoder(sample_rate=rate, nan_guard=True)
    o2.work([wb])
    assert len(frames) == len(o2.frames)
    for of, gf in zip(o2.frames, frames):
        assert_tfa_equal(of, gf)
    assert [f.packet() for f in frames if f.status == 0] == pkts


def test_long_row_pieces_decoded_in_parallel_ts33c(monkeypatch):
    """Rows of 128 KiB and more with four or more pieces: the closed middle pieces are decoded by
    k_decode_pieces, one wavefront each.  Frame records (status, limits from the histogram, packet,
    fields, end position) identical to the sequential walk (SDR_NO_PIECES=1) and equal to the oracle;
    junk bursts give error frames in the same slots; a frame table that is too small overflows alike."""
    rate = 1000000
    rng = np.random.default_rng(99)
    sym = []
    for i in range(7):
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
        if i % 3 == 1:
            for _ in range(25):
                sym += [(1, float(rng.integers(100, 1400))), (0, float(rng.integers(100, 1400)))]
        sym += [(0, float(rng.integers(2000, 30000)))]
    env = np.concatenate([synth.envelope(sym, rate, lead_in_us=3000.0), np.zeros(2000, np.uint8)]).astype(np.int8)
    assert env.size > 128 * 1024
    o = odec.TS33CStreamDecoder(sample_rate=rate)
    o.work([env])
    recs = {}
    for mode in ("pieces", "walk"):
        if mode == "walk":
            monkeypatch.setenv("SDR_NO_PIECES", "1")
        for max_frames in (64, 5):
            with chain.Chain(rate, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=env.size,
                             max_frames=max_frames) as c:
                try:
                    c.process_bits(env)
                    frames = [_copy(f) for f in c.fetch_frames()]
                    err = None
                except _lib.SdrError as ex:      # the overflow is reported when the frames are fetched
                    err, frames = str(ex), []
                recs[(mode, max_frames)] = (err, [bytes(bytearray(C.string_at(C.byref(f), C.sizeof(f)))) for f in frames],
                                            c.decoder_state())
            if max_frames == 64:
                assert err is None and len(frames) == len(o.frames) >= 6
                for of, gf in zip(o.frames, frames):
                    assert_ts33c_equal(of, gf)
    assert recs[("pieces", 64)] == recs[("walk", 64)]
    assert recs[("pieces", 5)] == recs[("walk", 5)] and recs[("pieces", 5)][0] is not None
