"""The sliding-window channel filter (k_frontend_win: decimation 1, real or complex taps, 10 outputs per lane) must
equal the oracle and the one-output-per-lane kernel word for word: same fmaf chain, oldest sample first.
SDR_FRONTEND (read by sdr_create) forces a kernel."""
import ctypes as C

import numpy as np
import pytest

from oracle import dsp as odsp
from tests.test_gpu_mfma import _chain, _iq, _same_words

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ntaps", [1, 2, 7, 15, 16, 17, 27, 31, 32, 33, 121, 333])
def test_win_fir_complex_out_bit_exact(ntaps):
    from sdr_amd import chain
    rng = np.random.default_rng(ntaps * 17)
    taps = rng.standard_normal(ntaps).astype(np.float32) / np.float32(max(1, ntaps) ** 0.5)
    for n in (40000 + 777, 1234, 5):          # 256-thread tiles + ragged tail; 64-thread tiles; shorter than the taps
        x = _iq(n, ntaps + n, ncap=3)
        got = {}
        for mode in ("win", "valu"):
            c = _chain(mode, 1e6, decim=1, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=3, max_samples=n)
            c.batch_process(x)
            got[mode] = c.fetch_demod().reshape(3, -1)
            c.close()
        assert _same_words(got["win"], got["valu"])
        for i in range(3):
            assert _same_words(got["win"][i], odsp.fir_ccf(x[i], taps, 1))


@pytest.mark.parametrize("demod", ["mag2", "quad"])
@pytest.mark.parametrize("keep_n", [1, 3, 100])
@pytest.mark.parametrize("stride_extra", [0, 1])
def test_win_demod_slicer_aligned_and_odd_strides(demod, keep_n, stride_extra):
    """even stride: 16-B DMA staging, captures whose tiles start on odd samples take the shifted tap table;
    odd stride: the guarded staging path.  Floats + sliced bytes vs the oracle chain."""
    from sdr_amd import _lib, chain
    taps = odsp.low_pass(1.0, 2e6, 500e3, 250e3, odsp.WIN_BLACKMAN) if demod == "quad" else odsp.low_pass(1.0, 2e6, 20e3, 40e3)
    n = 30000 + 301
    ncap = 3
    st = n + 1 + stride_extra      # n is odd: stride_extra 0 -> even stride (aligned), 1 -> odd
    flat = _iq(ncap * st, 11 + keep_n) * np.float32(0.5)
    x = np.stack([flat[i * st:i * st + n] for i in range(ncap)])
    gain = 3.0 if demod == "mag2" else 0.7
    lo, hi = (0.05, 0.15) if demod == "mag2" else (-0.3, 0.3)
    dm = chain.DEMOD_MAG2 if demod == "mag2" else chain.DEMOD_QUAD
    res = {}
    for mode in ("win", "valu"):
        c = _chain(mode, 1e6, decim=1, taps=taps, demod=dm, demod_gain=gain, thr=(lo, hi), keep_n=keep_n,
                   tap_demod=True, max_batch=ncap, max_samples=n)
        _lib.check(c._h, c._L.sdr_batch_process(c._h, C.c_void_p(flat.ctypes.data), n, st, ncap, 0, C.byref(c.out)))
        res[mode] = (c.fetch_demod().reshape(ncap, -1), c.fetch_bits().reshape(ncap, -1))
        c.close()
    assert _same_words(res["win"][0], res["valu"][0])
    np.testing.assert_array_equal(res["win"][1], res["valu"][1])
    for i in range(ncap):
        y = odsp.fir_ccf(x[i], taps, 1)
        v = odsp.mag2(y) if demod == "mag2" else odsp.quad_demod(y, 1.0)
        assert _same_words(res["win"][0][i], np.float32(gain) * v)
        b, _, _ = odsp.slice_bits(v, gain, lo, hi, keep_n)
        np.testing.assert_array_equal(res["win"][1][i], b)


def test_win_streaming_ragged_chunks_equal_one_shot():
    from sdr_amd import chain
    taps = odsp.low_pass(1.0, 2e6, 500e3, 250e3, odsp.WIN_BLACKMAN)     # 27 taps (TFA/decoder.grc)
    n = 60000
    x = _iq(n, 5) * np.float32(0.4)
    want_v = odsp.quad_demod(odsp.fir_ccf(x, taps, 1), 1.0)
    want_b, _, _ = odsp.slice_bits(want_v, 1.0, -0.2, 0.2, 3)
    c = _chain("win", 1e6, decim=1, taps=taps, demod=chain.DEMOD_QUAD, thr=(-0.2, 0.2), keep_n=3, tap_demod=True,
               max_samples=n)
    cuts = [0, 1, 2, 5, 1500, 1501, 4096, 4099, 7000, 40001, n]
    fl, bi = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        c.process_iq(x[a:b])
        fl.append(c.fetch_demod())
        bi.append(c.fetch_bits())
    c.close()
    assert _same_words(np.concatenate(fl), want_v)
    np.testing.assert_array_equal(np.concatenate(bi), want_b)


def test_win_nonfinite_sample_stays_local():
    """no 0*x products: an Inf sample spoils exactly the ntaps outputs whose window holds it (the shifted tap
    table of odd tile starts skips its dummy step instead of multiplying by zero)"""
    from sdr_amd import chain
    taps = odsp.low_pass(1.0, 2e6, 500e3, 250e3, odsp.WIN_BLACKMAN)
    n = 20000
    for pos in (0, 5000, 5001, 2559, 2560, n - 1):
        x = _iq(n, 9, ncap=2) * np.float32(0.3)
        x[1, pos] = np.inf
        c = _chain("win", 1e6, decim=1, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=2, max_samples=n)
        c.batch_process(x)
        y = c.fetch_demod().reshape(2, -1)
        c.close()
        bad = ~np.isfinite(y[1])
        assert bad.sum() == min(taps.size, n - pos) and bad[pos:pos + taps.size].all()
        assert np.isfinite(y[0]).all()


@pytest.mark.parametrize("ntaps,fc", [(1, 250e3), (2, -100e3), (16, 37e3), (17, 100e3), (27, -500e3), (121, 100e3)])
def test_win_xlating_complex_taps_bit_exact(ntaps, fc):
    """freq_xlating_fir_filter_ccc at decimation 1 (complex band-pass taps, then the NCO): the window kernel's
    complex-tap form against the one-output-per-lane kernel and the oracle, complex outputs word for word;
    even and odd capture strides (aligned DMA staging with the shifted tap table / guarded staging)."""
    from sdr_amd import _lib, chain
    rng = np.random.default_rng(ntaps * 13 + 1)
    taps = rng.standard_normal(ntaps).astype(np.float32) / np.float32(max(1, ntaps) ** 0.5)
    fs = 1e6
    for n, extra in ((30000 + 301, 0), (30000 + 301, 1), (777, 0), (3, 0)):
        ncap = 3
        st = n + 1 + extra
        flat = _iq(ncap * st, ntaps + n) * np.float32(0.5)
        got = {}
        for mode in ("win", "valu"):
            c = _chain(mode, fs, decim=1, center_freq=fc, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True,
                       max_batch=ncap, max_samples=n)
            _lib.check(c._h, c._L.sdr_batch_process(c._h, C.c_void_p(flat.ctypes.data), n, st, ncap, 0, C.byref(c.out)))
            got[mode] = c.fetch_demod().reshape(ncap, -1)
            c.close()
        assert _same_words(got["win"], got["valu"])
        for i in range(ncap):
            assert _same_words(got["win"][i], odsp.freq_xlating_fir(flat[i * st:i * st + n], taps, fc, fs, 1))


@pytest.mark.parametrize("demod,keep_n", [("mag2", 1), ("mag2", 50), ("quad", 1), ("quad", 3)])
def test_win_xlating_demod_slicer_and_streaming(demod, keep_n):
    """The reference-wired TS33C graph with a carrier offset (shape R, xlating variant): 121
    translated taps at decimation 1 -> demod -> threshold -> slicer -> keep 1-in-n; batch and ragged streaming
    chunks, floats and bytes against the oracle blocks."""
    from sdr_amd import chain
    fs, fc = 1e6, 100e3
    taps = odsp.low_pass(1.0, fs, 20e3, 40e3)
    n = 50000 + 13
    x = _iq(n, 77 + keep_n) * np.float32(0.4)
    y = odsp.freq_xlating_fir(x, taps, fc, fs, 1)
    gain = 2.0 if demod == "mag2" else 0.9
    lo, hi = (0.05, 0.15) if demod == "mag2" else (-0.3, 0.3)
    v = odsp.mag2(y) if demod == "mag2" else odsp.quad_demod(y, 1.0)
    wb, _, _ = odsp.slice_bits(v, gain, lo, hi, keep_n)
    dm = chain.DEMOD_MAG2 if demod == "mag2" else chain.DEMOD_QUAD
    for tap in (True, False):                      # without the float tap interior tiles take the packed-store path
        c = _chain("win", fs, decim=1, center_freq=fc, taps=taps, demod=dm, demod_gain=gain, thr=(lo, hi), keep_n=keep_n,
                   tap_demod=tap, max_samples=n)
        c.batch_process(x[None, :])
        if tap:
            assert _same_words(c.fetch_demod(), np.float32(gain) * v)
        np.testing.assert_array_equal(c.fetch_bits(), wb)
        c.close()
    c = _chain("win", fs, decim=1, center_freq=fc, taps=taps, demod=dm, demod_gain=gain, thr=(lo, hi), keep_n=keep_n,
               tap_demod=True, max_samples=n)
    cuts = [0, 1, 2, 7, 2559, 2560, 2561, 9000, 9001, 40001, n]
    fl, bi = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        c.process_iq(x[a:b])
        fl.append(c.fetch_demod())
        bi.append(c.fetch_bits())
    c.close()
    assert _same_words(np.concatenate(fl), np.float32(gain) * v)
    np.testing.assert_array_equal(np.concatenate(bi), wb)


def _quad_bits_oracle(x, taps, gain, lo, hi, keep_n=1):
    y = odsp.fir_ccf(x, taps, 1)
    v = odsp.quad_demod(y, 1.0)
    return odsp.slice_bits(v, gain, lo, hi, keep_n)[0]


@pytest.mark.parametrize("gain,lo,hi", [(2.6526, 0.1, 0.4), (0.7, -0.3, 0.3), (1.0, -1.2, -0.4), (5.0, 0.25, 0.25),
                                        (0.31, -0.9, 0.95), (1.0, 0.0, 0.0), (3.0, -1e-3, 2e-3)])
def test_quad_threshold_only_path_equals_full_arctangent(gain, lo, hi):
    """Round 3: a discriminator chain that only slices (no demod tap) classifies gain*fast_atan2f(ti, tr) against the
    thresholds by cross-product signs outside a guard band (frontend_win.h) and evaluates the arctangent only inside
    it.  Sliced bytes must equal the chain that keeps the floats (full arctangent everywhere) and the oracle, on
    noise (angles uniform: every direction incl. the band gets hit), an FSK-like tone pair, zeros and tiny samples."""
    from sdr_amd import chain
    rng = np.random.default_rng(int(1000 * gain) + 7)
    taps = odsp.low_pass(1.0, 2e6, 500e3, 250e3, odsp.WIN_BLACKMAN)
    n, ncap = 60000 + 123, 4
    x = np.zeros((ncap, n), np.complex64)
    x[0] = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    f = np.where((np.arange(n) // 52) % 2 == 0, 0.02, -0.02) * (1 + 0.2 * rng.standard_normal(n))
    x[1] = (0.7 * np.exp(2j * np.pi * np.cumsum(f))).astype(np.complex64) + 0.02 * x[0]
    x[2, 1000:30000] = x[0, 1000:30000] * np.float32(1e-20)           # zero length and denormal products
    x[3] = x[0] * np.float32(1e15)
    x[3, 5000] = np.inf
    x[3, 9000] = np.nan
    res = {}
    for tap in (False, True):
        with chain.Chain(2e6, decim=1, taps=taps, demod=chain.DEMOD_QUAD, demod_gain=gain, thr=(lo, hi), tap_demod=tap,
                         max_batch=ncap, max_samples=n) as c:
            c.batch_process(x)
            res[tap] = c.fetch_bits().reshape(ncap, -1)
    np.testing.assert_array_equal(res[False], res[True])
    for i in range(3):
        np.testing.assert_array_equal(res[False][i], _quad_bits_oracle(x[i], taps, np.float32(gain), lo, hi))
    # streaming in ragged chunks == one shot
    with chain.Chain(2e6, decim=1, taps=taps, demod=chain.DEMOD_QUAD, demod_gain=gain, thr=(lo, hi), max_samples=n) as c:
        parts = []
        for a, b in zip([0, 7, 3000, 3001, 40000], [7, 3000, 3001, 40000, n]):
            c.process_iq(x[1][a:b])
            parts.append(c.fetch_bits())
    np.testing.assert_array_equal(np.concatenate(parts), _quad_bits_oracle(x[1], taps, np.float32(gain), lo, hi)[:sum(p.size for p in parts)])


def test_quad_threshold_only_path_exactly_at_the_thresholds():
    """Samples whose arctangent lies ON a threshold and within a few ulp / 1e-7 .. 1e-3 rad of it, both thresholds,
    both sides, all magnitudes: with y[n-1] = 1 the discriminator's argument is the sample itself.  The 9-tap filter
    [1, 0, ...] passes the samples through exactly.  Bytes equal the oracle's."""
    from sdr_amd import chain
    gain, lo, hi = np.float32(2.6526), 0.1, 0.4
    taps = np.zeros(9, np.float32)
    taps[0] = 1.0
    offs = np.concatenate([[0.0], np.logspace(-9, -2.5, 120), -np.logspace(-9, -2.5, 120)])
    ang = np.concatenate([hi / float(gain) + offs, lo / float(gain) + offs, -hi / float(gain) + offs, np.pi - np.abs(offs), -np.pi + np.abs(offs)])
    mags = np.array([1.0, 1e-3, 37.5, 1e-12, 1e12])
    pts = (mags[:, None] * np.exp(1j * ang[None, :])).reshape(-1)
    x = np.ones(2 * pts.size + 2, np.complex64)
    x[1:-1:2] = pts.astype(np.complex64)          # 1, p0, 1, p1, ...: arguments p_k and conj(p_k) alternate
    want = _quad_bits_oracle(x, taps, gain, lo, hi)
    with chain.Chain(2e6, decim=1, taps=taps, demod=chain.DEMOD_QUAD, demod_gain=float(gain), thr=(lo, hi), max_samples=x.size) as c:
        c.batch_process(x[None, :])
        got = c.fetch_bits()
    np.testing.assert_array_equal(got, want)
    assert 0 < want.sum() < want.size
