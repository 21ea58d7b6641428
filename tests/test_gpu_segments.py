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
