"""One long capture cut into contiguous segments with FIR-history overlap
(SURVEY.md 8e): every segment runs on a handle positioned with sdr_seek, the
hysteresis carries are resolved across segments and the stitched sliced stream
must equal -- byte for byte -- the oracle's single pass over the whole capture,
and decode to the same frames."""
import numpy as np
import pytest

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import chain, segments, synth
from tests.test_host_decode import assert_ts33c_equal

pytestmark = pytest.mark.gpu

FS = 1e6


def _taps():
    return chain.low_pass(1.0, FS, 20e3, 40e3, chain.WIN_HAMMING)


def _run_segments(iq, bounds, **kw):
    n = int(iq.shape[0])
    bits, infos = [], []
    with chain.Chain(FS, max_samples=n, **kw) as c:
        for a, b in bounds:
            infos.append(segments.run_segment(c, iq, a, b))
            bits.append(c.fetch_bits())
    return bits, infos


@pytest.mark.parametrize("D,keep,fc,world,align", [
    (50, 1, 100e3, 4, 50), (50, 1, 0.0, 8, 2), (1, 50, 0.0, 3, 2), (5, 10, -37e3, 5, 1), (64, 1, 0.0, 8, 1)])
def test_segments_equal_single_pass(D, keep, fc, world, align):
    n = (1 << 19) + 1234
    iq, pkt, _ = synth.ts33c_capture(n, FS, seed=91, f_off=fc)
    taps = _taps()
    bounds = segments.segment_bounds(n, world, align)
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    bits, infos = _run_segments(iq, bounds, decim=D, center_freq=fc, taps=taps, keep_n=keep)
    want = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, keep, fc=fc, fs=FS)
    got = segments.stitch_segments(bits, infos, 0)
    assert got.size == want.size
    assert np.array_equal(got, want)
    # the bookkeeping the collective relies on
    off = 0
    for b, i in zip(bits, infos):
        assert i.n_bits == b.size and i.first_bit_index == off and i.carry_in == 0
        off += b.size
    assert want.sum() > 100


def test_segment_cut_inside_a_pulse_needs_the_carry():
    """A boundary inside a long pulse whose magnitude sits between the two
    thresholds: the segment's first bytes are undetermined and must take the
    previous segment's state."""
    n = 1 << 16
    env = np.zeros(n, np.float32)
    env[1000:20000] = 1.0
    env[20000:50000] = 0.3  # between lo=.05 and hi=.15 after mag^2 (0.09): holds the state
    iq = (env * np.exp(1j * 0.3)).astype(np.complex64)
    bounds = [(0, 32768), (32768, n)]
    bits, infos = _run_segments(iq, bounds, decim=8, taps=[1.0], keep_n=1)
    assert infos[1].n_undetermined > 1000 and infos[0].last_state == 1
    want = odsp.ook_chain(iq, np.ones(1, np.float32), 8, 1.0, 0.05, 0.15, 1)
    naive = np.concatenate(bits)
    assert not np.array_equal(naive, want)
    assert np.array_equal(segments.stitch_segments(bits, infos, 0), want)


def test_segments_quad_demod_and_decode():
    """Discriminator chain (history = previous output) in segments, and the
    stitched TS33C stream decoded by one decoder handle fed from device memory."""
    import torch
    n = 1 << 20
    iq, pkt, _ = synth.ts33c_capture(n, FS, seed=17)
    taps = _taps()
    D = 50
    bounds = segments.segment_bounds(n, 8, D)
    x = torch.from_numpy(iq.view(np.float32).reshape(n, 2)).to("cuda:0")
    bits, infos = _run_segments(x, bounds, decim=D, taps=taps, keep_n=1)
    want = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, 1)
    got = segments.stitch_segments(bits, infos, 0)
    assert np.array_equal(got, want)
    o = odec.TS33CStreamDecoder(sample_rate=20000)
    o.work([want])
    with chain.Chain(20000.0, decoder=chain.DEC_TS33C, max_samples=got.size, max_frames=64) as dec:
        dec.process_bits(torch.from_numpy(got).to("cuda:0"))
        frames = dec.fetch_frames()
    assert len(frames) == len(o.frames) >= 2
    for of, gf in zip(o.frames, frames):
        assert_ts33c_equal(of, gf)
    # quadrature demod: floats of the segments == one pass
    rng = np.random.default_rng(2)
    m = 40000
    z = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
    t27 = chain.low_pass(1.0, 2e6, 500e3, 250e3, chain.WIN_BLACKMAN)
    with chain.Chain(2e6, decim=2, taps=t27, demod=chain.DEMOD_QUAD, demod_gain=1.5, thr=(-0.2, 0.2), tap_demod=True,
                     max_samples=m) as c:
        c.batch_process(z[None, :])
        one = c.fetch_demod().copy()
        one_bits = c.fetch_bits().copy()
        parts, pb, infos = [], [], []
        for a, b in [(0, 9999), (9999, 10001), (10001, 33333), (33333, m)]:
            infos.append(segments.run_segment(c, z, a, b))
            parts.append(c.fetch_demod().copy())
            pb.append(c.fetch_bits().copy())
    assert np.array_equal(np.concatenate(parts).view(np.uint32), one.view(np.uint32))
    assert np.array_equal(segments.stitch_segments(pb, infos, 0), one_bits)


def test_segments_of_multi_stage_chains():
    """Chains with more than one memory are positioned by a warm-up over a longer halo (sdr_history_len grows
    accordingly): the mumbi FS300 two-filter graph (mumbi/FS300.grc:3157-3242: 97-tap LPF -> 329-tap 1/10
    resampler -> mag^2 -> keep 1-in-20) and the TFA analyzer2 graph (TFA/analyzer2.grc:3151-3232: rotate +500 kHz
    -> mag^2 -> 7-tap float LPF) cut into ragged segments: floats and stitched bytes == the oracle's one pass."""
    from tests.test_gpu_chain import _assert_float_parity
    fs = 2e6
    rng = np.random.default_rng(78)
    # --- prefilter + decimating second stage
    sym = synth.mumbi_symbols(bytes([0x12, 0x34, 0x56, 0x78, 0x80]), repeats=4)
    iq = synth.ook_iq(synth.envelope(sym, fs, lead_in_us=9000.0), fs, amp=0.9, sigma=0.02, rng=rng)
    n = iq.size
    lpf = chain.low_pass(1.0, fs, 100e3, 50e3)
    rs = chain.resampler_taps(1, 10)
    wv = odsp.mag2(odsp.fir_ccf(odsp.fir_ccf(iq, lpf, 1), rs, 10))[:n // 10]
    wb, _, _ = odsp.slice_bits(wv, 1.0, 0.1, 0.4, 20)
    cuts = [0, 1237, 1240, 50001, n // 2 + 3, n - 777, n]
    with chain.Chain(fs, pre_taps=lpf, pre_decim=1, decim=10, taps=rs, demod=chain.DEMOD_MAG2, thr=(0.1, 0.4),
                     keep_n=20, tap_demod=True, max_samples=n) as c:
        H = c.history_len()
        assert H >= 96 + 328 + 10
        with pytest.raises(_lib_err()):
            c.seek(100000, iq[100000 - H + 1:100000])      # one sample short
        # segments in REVERSE order on one handle: nothing may leak from one seek to the next
        res = {}
        for a, b in reversed(list(zip(cuts[:-1], cuts[1:]))):
            info = segments.run_segment(c, iq, a, b)
            res[a] = (info, c.fetch_bits().copy(), c.fetch_demod().copy())
    order = sorted(res)
    assert _assert_float_parity(np.concatenate([res[a][2] for a in order]), wv) == 0
    got = segments.stitch_segments([res[a][1] for a in order], [res[a][0] for a in order], 0)
    assert np.array_equal(got, wb[:got.size]) and got.size == wb.size and wb.sum() > 50
    # --- post-demod float filter behind a rotated input
    sym = []
    for _ in range(2):
        pk = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=21.5, humidity=int(rng.integers(0, 100)))
        sym += synth.tfa_symbols(synth.tfa_bits(pk), rng, reset_gap_us=900.0)
    env = np.concatenate([synth.envelope(sym, fs, lead_in_us=800.0), np.zeros(2500, np.uint8)])
    iq = synth.ook_iq(env, fs, f_off=500e3, amp=0.9, sigma=0.01, rng=rng)
    n = iq.size
    post = chain.low_pass(1.0, fs, 500e3, 1e6, chain.WIN_BLACKMAN)
    wv = odsp.fir_fff(odsp.mag2(odsp.freq_xlating_fir(iq, [1.0], 500e3, fs, 1)), post, 1)
    wb, _, _ = odsp.slice_bits(wv, 1.0, 0.2, 0.4, 1)
    bounds = [(0, 5), (5, 6), (6, 4099), (4099, n // 2), (n // 2, n)]
    with chain.Chain(fs, decim=1, center_freq=500e3, demod=chain.DEMOD_MAG2, thr=(0.2, 0.4), keep_n=1, post_taps=post,
                     tap_demod=True, max_samples=n) as c:
        infos, bits, vs = [], [], []
        for a, b in bounds:
            infos.append(segments.run_segment(c, iq, a, b))
            bits.append(c.fetch_bits().copy())
            vs.append(c.fetch_demod().copy())
    assert _assert_float_parity(np.concatenate(vs), wv) == 0
    assert np.array_equal(segments.stitch_segments(bits, infos, 0), wb) and wb.sum() > 1000


def _lib_err():
    from sdr_amd import _lib
    return _lib.SdrError


def test_seek_argument_checks():
    from sdr_amd import _lib
    with chain.Chain(FS, decim=10, taps=_taps(), max_samples=4096) as c:
        assert c.history_len() == 60 + 20  # 61 taps at 1 Msps, + 2*D
        with pytest.raises(_lib.SdrError):
            c.seek(1000, np.zeros(10, np.complex64))   # halo too short
        c.seek(5, np.zeros(5, np.complex64))           # stream start: 5 samples suffice
    with chain.Chain(FS, decim=10, taps=_taps(), max_samples=4096, squelch=(-40.0, 1e-4)) as c:
        with pytest.raises(_lib.SdrError):
            c.seek(0)


# ---- frames-only exchange (sdr_segment_decode + sdr_process_runs; north_star: RCCL carries frames and counts only)
def _key(f):
    return (int(f.end_pos), int(f.status), f.packet(), int(f.nbits), f.lim[0], f.lim[1])


def _okey(of):
    from tests.test_segments_frames import to_frame
    return _key(to_frame(of))


def _long_ts33c_capture(seed, n_pk=7, held=True):
    """several different packets, a long quiet stretch, and (held) a stretch that sits between the two thresholds
    entered from a pulse, so that a segment starting inside it is undetermined until the next event"""
    rng = np.random.default_rng(seed)
    envs, amps = [], []
    sym = []
    for k in range(n_pk):
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
        if k == 2:
            sym += [(0, 60000.0)]
    env = synth.envelope(sym, FS, lead_in_us=4000.0).astype(np.float32)
    env = np.concatenate([env, np.zeros(5000, np.float32)])
    iq = synth.ook_iq((env > 0).astype(np.uint8), FS, amp=0.7, sigma=0.02, rng=rng)
    hold = None
    if held:
        # in the quiet stretch behind packet 3: 3 ms of carrier, then 30 ms at |x|^2 = 0.09 (between .05 and .15)
        edges = np.nonzero(np.diff((env > 0).astype(np.int8)))[0]
        gaps = np.diff(edges)
        g = int(np.argmax(gaps))
        a = int(edges[g]) + 8000
        iq[a:a + 3000] = 0.7
        iq[a + 3000:a + 33000] = 0.3
        hold = (a + 3000, a + 33000)
    return iq, hold


@pytest.mark.parametrize("world,D,keep", [(3, 50, 1), (8, 50, 1), (5, 10, 5), (8, 1, 50)])
def test_frames_only_exchange_equals_single_pass(world, D, keep):
    iq, hold = _long_ts33c_capture(world * 7 + D)
    n = iq.size
    taps = _taps()
    want_bits = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, keep)
    o = odec.TS33CStreamDecoder(sample_rate=FS / D / keep)
    o.work([want_bits])
    want = [_okey(f) for f in o.frames]
    assert len(want) >= 6
    al = D * keep
    bounds = segments.segment_bounds(n, world, al)
    # one boundary goes inside the held stretch, one into the middle of a packet (the first frame's end - 30 ms)
    cut_hold = (hold[0] + 9000) // al * al
    cut_pkt = max(al, (want[0][0] * al - 30000) // al * al)
    inner = sorted(set([b for _, b in bounds[:-1]][:world - 3] + [cut_hold, cut_pkt]))
    cuts = [0] + inner + [n]
    bounds = list(zip(cuts[:-1], cuts[1:]))
    with chain.Chain(FS, decim=D, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, max_samples=n, max_frames=32) as c:
        got, stats = segments.frames_only(c, iq, bounds)
        assert not stats["fallback"]
        assert [_key(f) for f in got] == want
        assert stats["local_frames"] >= 1 and stats["stitched_frames"] >= 2
        assert stats["table_entries"] * 4 < 64 * 1024
        # the same handle goes on as a streaming decoder afterwards
        c.seek(0)
        c.process_iq(iq)
        assert [_key(f) for f in c.fetch_frames()] == want
        # tables that do not fit are flagged, not truncated
        got2, stats2 = segments.frames_only(c, iq, bounds, cap=8)
        assert got2 is None and stats2["fallback"]


def test_frames_only_exchange_tfa():
    """TFA analyzer2 shape at 2 Msps (reset_limit 500 us, ~250 widths per packet): segments cut packets in the
    sync run and in the data; the segment's own pass meets the reference's latent exceptions on the cut packets
    (status codes here) and the keep window drops them."""
    fs = 2e6
    rng = np.random.default_rng(41)
    sym = []
    for _ in range(6):
        pk = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10.0,
                              humidity=int(rng.integers(0, 100)))
        sym += synth.tfa_symbols(synth.tfa_bits(pk), rng, reset_gap_us=float(rng.integers(700, 3000)))
    env = np.concatenate([synth.envelope(sym, fs, lead_in_us=800.0), np.zeros(2500, np.uint8)])
    iq = synth.ook_iq(env, fs, amp=0.9, sigma=0.01, rng=rng)
    n = iq.size
    want_bits = odsp.ook_chain(iq, np.ones(1, np.float32), 1, 1.0, 0.2, 0.4, 1)
    o = odec.TFAStreamDecoder(sample_rate=2000000)
    o.work([want_bits])
    want = [_okey(f) for f in o.frames]
    assert len(want) == 6 and all(k[1] == 0 for k in want)
    for world in (2, 5, 8):
        inner = sorted(int(v) for v in rng.integers(1, n, world - 1))
        cuts = [0] + inner + [n]
        with chain.Chain(fs, decim=1, taps=[1.0], thr=(0.2, 0.4), keep_n=1, decoder=chain.DEC_TFA, max_samples=n,
                         max_frames=32) as c:
            got, stats = segments.frames_only(c, iq, list(zip(cuts[:-1], cuts[1:])))
            assert not stats["fallback"]
            assert [_key(f) for f in got] == want, cuts


def test_process_runs_is_one_work_call():
    """sdr_process_runs on the run table of a stream == sdr_process_bits on its bytes, state carried over calls"""
    rng = np.random.default_rng(9)
    from tests.test_segments_frames import _ts33c_stream
    bits = _ts33c_stream(rng, 4, held=True)
    pos = np.concatenate([[0], np.nonzero(np.diff(bits))[0] + 1, [bits.size]]).astype(np.int32)
    with chain.Chain(20000.0, decoder=chain.DEC_TS33C, max_samples=bits.size, max_frames=16) as a, \
            chain.Chain(20000.0, decoder=chain.DEC_TS33C, max_samples=bits.size, max_frames=16) as b:
        for _ in range(2):   # second call: pending symbols and previousSymbolType of the first are in play
            a.process_bits(bits)
            b.process_runs(pos, int(bits[0]), int(bits[-1]))
            fa, fb = a.fetch_frames(), b.fetch_frames()
            assert len(fa) >= 3 and [_key(f) for f in fa] == [_key(f) for f in fb]
            assert a.decoder_state() == b.decoder_state()
            assert np.array_equal(a.fetch_pending_symbols(), b.fetch_pending_symbols())
        with pytest.raises(_lib_err()):
            b.process_runs(np.array([0, 5, 5, 9], np.int32), 0, 0)
        with pytest.raises(_lib_err()):
            b.process_runs(np.array([0, 9], np.int32), 0, 0)      # shorter than reset_limit


@pytest.mark.parametrize("fmt,fc", [("fc32", 100e3), ("cu8", 0.0)])
def test_frames_only_exchange_xlating_and_cu8(fmt, fc):
    """the same with a translated channel (NCO phase set by sdr_seek) and with rtl-sdr bytes as input"""
    from oracle import dsp as odsp
    rng = np.random.default_rng(5)
    sym = []
    for k in range(5):
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
    env = np.concatenate([synth.envelope(sym, FS, lead_in_us=3000.0), np.zeros(4000, np.uint8)])
    iq = synth.ook_iq(env, FS, f_off=fc, amp=0.7, sigma=0.02, rng=rng)
    n = iq.size
    taps = _taps()
    D = 50
    kw = dict(decim=D, taps=taps, keep_n=1, center_freq=fc, decoder=chain.DEC_TS33C, max_samples=n, max_frames=32)
    if fmt == "cu8":
        u8 = np.clip(np.round(iq.view(np.float32) * 128.0 + 127.4), 0, 255).astype(np.uint8).reshape(n, 2)
        x = u8
        xf = ((u8.astype(np.float32) - np.float32(127.4)) / np.float32(128.0)).view(np.complex64).reshape(n)
        kw["input_format"] = "cu8"
    else:
        x, xf = iq, iq
    want_bits = odsp.ook_chain(xf, taps, D, 1.0, 0.05, 0.15, 1, fc=fc, fs=FS)
    o = odec.TS33CStreamDecoder(sample_rate=FS / D)
    o.work([want_bits])
    want = [_okey(f) for f in o.frames]
    assert len(want) >= 4
    inner = sorted(int(v) // D * D for v in rng.integers(D, n - D, 5))
    cuts = [0] + inner + [n]
    with chain.Chain(FS, **kw) as c:
        got, stats = segments.frames_only(c, x, list(zip(cuts[:-1], cuts[1:])))
        assert not stats["fallback"]
        assert [_key(f) for f in got] == want, cuts
