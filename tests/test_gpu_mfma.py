"""The matrix-core channel filter (k_frontend_mfma) must equal the oracle and the
LDS/VALU kernel word for word: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain,
so the banded-Toeplitz FIR keeps the oldest-sample-first contract exactly.
SDR_FRONTEND (read by sdr_create) forces either kernel."""
import os

import numpy as np
import pytest

from oracle import dsp as odsp

pytestmark = pytest.mark.gpu


def _chain(mode, *a, **k):
    from sdr_amd import chain
    old = os.environ.get("SDR_FRONTEND")
    os.environ["SDR_FRONTEND"] = mode
    try:
        return chain.Chain(*a, **k)
    finally:
        if old is None:
            del os.environ["SDR_FRONTEND"]
        else:
            os.environ["SDR_FRONTEND"] = old


def _iq(n, seed, ncap=None):
    rng = np.random.default_rng(seed)
    shape = (n,) if ncap is None else (ncap, n)
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


def _same_words(a, b):
    a = np.ascontiguousarray(a).view(np.uint32)
    b = np.ascontiguousarray(b).view(np.uint32)
    return a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("D,ntaps", [(1, 1), (1, 2), (1, 27), (1, 121), (2, 61), (3, 40), (4, 131), (5, 7), (8, 121),
                                     (10, 329), (16, 300)])
def test_mfma_fir_complex_out_bit_exact(D, ntaps):
    from sdr_amd import chain
    rng = np.random.default_rng(ntaps * 31 + D)
    taps = rng.standard_normal(ntaps).astype(np.float32) / np.float32(max(1, ntaps) ** 0.5)
    n = 3 * 1024 * D + 777          # several tiles + a ragged tail
    x = _iq(n, D + ntaps, ncap=3)
    got = {}
    for mode in ("mfma", "valu"):
        c = _chain(mode, 1e6, decim=D, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=3, max_samples=n)
        c.batch_process(x)
        got[mode] = c.fetch_demod().reshape(3, -1)
        c.close()
    assert _same_words(got["mfma"], got["valu"])
    for i in range(3):
        assert _same_words(got["mfma"][i], odsp.fir_ccf(x[i], taps, D))


@pytest.mark.parametrize("demod", ["mag2", "quad"])
@pytest.mark.parametrize("D,keep_n", [(1, 100), (4, 3), (1, 1)])
def test_mfma_demod_slicer_and_unaligned_captures(demod, D, keep_n):
    """odd capture stride (no 16-B staging), floats + bits vs the oracle chain"""
    import ctypes as C
    from sdr_amd import _lib, chain
    taps = odsp.low_pass(1.0, 2e6, 20e3, 40e3)      # 121 taps
    n = 2 * 1024 * D + 301
    ncap = 3
    flat = _iq(ncap * (n + 1), 11 * D + keep_n) * np.float32(0.5)
    x = np.stack([flat[i * (n + 1):i * (n + 1) + n] for i in range(ncap)])
    gain = 3.0 if demod == "mag2" else 0.7
    lo, hi = (0.05, 0.15) if demod == "mag2" else (-0.3, 0.3)
    dm = chain.DEMOD_MAG2 if demod == "mag2" else chain.DEMOD_QUAD
    res = {}
    for mode in ("mfma", "valu"):
        c = _chain(mode, 1e6, decim=D, taps=taps, demod=dm, demod_gain=gain, thr=(lo, hi), keep_n=keep_n,
                   tap_demod=True, max_batch=ncap, max_samples=n)
        # odd stride between captures (the library keeps the stride in its staging copy)
        _lib.check(c._h, c._L.sdr_batch_process(c._h, C.c_void_p(flat.ctypes.data), n, n + 1, ncap, 0, C.byref(c.out)))
        res[mode] = (c.fetch_demod().reshape(ncap, -1), c.fetch_bits().reshape(ncap, -1))
        c.close()
    assert _same_words(res["mfma"][0], res["valu"][0])
    np.testing.assert_array_equal(res["mfma"][1], res["valu"][1])
    for i in range(ncap):
        y = odsp.fir_ccf(x[i], taps, D)
        v = odsp.mag2(y) if demod == "mag2" else odsp.quad_demod(y, 1.0)
        assert _same_words(res["mfma"][0][i], np.float32(gain) * v)
        b, _, _ = odsp.slice_bits(v, gain, lo, hi, keep_n)
        np.testing.assert_array_equal(res["mfma"][1][i], b)


def test_mfma_streaming_ragged_chunks_equal_one_shot():
    from sdr_amd import chain
    taps = odsp.low_pass(1.0, 2e6, 100e3, 50e3)     # 97 taps (mumbi stage 1)
    n = 9000
    x = _iq(n, 5) * np.float32(0.4)
    want_v = odsp.quad_demod(odsp.fir_ccf(x, taps, 2), 1.0)
    want_b, _, _ = odsp.slice_bits(want_v, 1.0, -0.2, 0.2, 3)
    c = _chain("mfma", 1e6, decim=2, taps=taps, demod=chain.DEMOD_QUAD, thr=(-0.2, 0.2), keep_n=3, tap_demod=True,
               max_samples=n)
    cuts = [0, 1, 2, 5, 1500, 1501, 4096, 4099, 7000, n]
    fl, bi = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        c.process_iq(x[a:b])
        fl.append(c.fetch_demod())
        bi.append(c.fetch_bits())
    c.close()
    assert _same_words(np.concatenate(fl), want_v)
    np.testing.assert_array_equal(np.concatenate(bi), want_b)


def test_auto_mode_equals_both_kernels_on_the_r_shape():
    """the reference-faithful R shape (121 taps, D = 1, keep 1-in-100): whichever
    kernel auto mode picks, the sliced stream is the same"""
    from sdr_amd import chain
    taps = odsp.low_pass(1.0, 2e6, 20e3, 40e3)
    x = _iq(1 << 16, 3, ncap=2) * np.float32(0.3)
    bits = {}
    for mode in ("auto", "valu", "mfma"):
        c = _chain(mode, 1e6, decim=1, taps=taps, demod_gain=30.0, keep_n=100, max_batch=2, max_samples=1 << 16)
        c.batch_process(x)
        bits[mode] = c.fetch_bits()
        c.close()
    np.testing.assert_array_equal(bits["auto"], bits["valu"])
    np.testing.assert_array_equal(bits["mfma"], bits["valu"])
    assert bits["valu"].any() and not bits["valu"].all()


@pytest.mark.parametrize("D,ntaps", [(1, 27), (2, 61), (4, 131), (3, 200)])
def test_mfma_two_team_kernel_batch_stream_and_fallback(D, ntaps, monkeypatch):
    """k_frontend_mfma2 (two teams per workgroup in antiphase, tiles chained, complex output) against the oracle
    and against k_frontend_mfma (SDR_NO_MFMA2, read at create): aligned and unaligned capture strides, a batch,
    ragged streaming chunks through the FIR history, tile counts that leave one team a tile short."""
    import ctypes as C
    from sdr_amd import _lib, chain
    rng = np.random.default_rng(D * 1000 + ntaps)
    taps = rng.standard_normal(ntaps).astype(np.float32) / np.float32(ntaps ** 0.5)
    for tiles_x, extra in ((5, 0), (2, 1), (9, 0)):
        n = tiles_x * 1024 * D + 333
        ncap = 2
        st = n + (n & 1) + extra                      # even stride: 16-B loads; odd: guarded, sample by sample
        flat = _iq(ncap * st, n + extra)
        got = {}
        for two in (True, False):
            if two:
                monkeypatch.delenv("SDR_NO_MFMA2", raising=False)
            else:
                monkeypatch.setenv("SDR_NO_MFMA2", "1")
            c = _chain("mfma", 1e6, decim=D, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=ncap, max_samples=n)
            _lib.check(c._h, c._L.sdr_batch_process(c._h, C.c_void_p(flat.ctypes.data), n, st, ncap, 0, C.byref(c.out)))
            got[two] = c.fetch_demod().reshape(ncap, -1)
            c.close()
        assert _same_words(got[True], got[False])
        for i in range(ncap):
            assert _same_words(got[True][i], odsp.fir_ccf(flat[i * st:i * st + n], taps, D))
    monkeypatch.delenv("SDR_NO_MFMA2", raising=False)
    n = 7 * 1024 * D + 5
    x = _iq(n, 3 * D + ntaps)
    want = odsp.fir_ccf(x, taps, D)
    c = _chain("mfma", 1e6, decim=D, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_samples=n)
    cuts = [0, 1, 2 * 1024 * D + 1, 2 * 1024 * D + 2, 5 * 1024 * D + 7, n]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        c.process_iq(x[a:b])
        parts.append(c.fetch_demod())
    c.close()
    assert _same_words(np.concatenate(parts), want)


def _same_or_both_nan(a, b):
    """word for word where finite; NaN where NaN (payload and sign of a NaN are not part of the contract); Inf == Inf"""
    a = np.ascontiguousarray(a).view(np.float32).ravel()
    b = np.ascontiguousarray(b).view(np.float32).ravel()
    if a.shape != b.shape:
        return False
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and np.array_equal(a[~nan].view(np.uint32), b[~nan].view(np.uint32))


@pytest.mark.parametrize("two_team", [False, True])
@pytest.mark.parametrize("D,ntaps", [(1, 27), (4, 131), (2, 61)])
@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan])
def test_mfma_nonfinite_sample_stays_local(D, ntaps, bad, two_team, monkeypatch):
    """The zero-padded products of the banded-Toeplitz form would turn an Inf / NaN sample into NaN for the <= 15
    outputs of its 16-output group whose window does not hold it (round-2 limitation).  A wave that ends with a
    non-finite accumulator recomputes its outputs over the taps alone, so exactly the outputs whose window holds the
    sample are non-finite -- as in the oracle (one fmaf per tap) and in k_frontend_win -- and every finite word
    still equals the oracle's.  Both matrix-core front ends, samples at tile / group edges and in the last tile."""
    from sdr_amd import chain
    if two_team:
        monkeypatch.delenv("SDR_NO_MFMA2", raising=False)
    else:
        monkeypatch.setenv("SDR_NO_MFMA2", "1")
    rng = np.random.default_rng(ntaps + D)
    taps = (rng.standard_normal(ntaps).astype(np.float32) / np.float32(ntaps ** 0.5))
    taps[taps == 0] = np.float32(0.01)
    n = 5 * 1024 * D + 333
    for pos in (0, 7, 16 * D - 1, 16 * D, 1024 * D - 1, 1024 * D, 2 * 1024 * D + 16 * D * 7 + 3, n - ntaps - 2, n - 1):
        x = _iq(n, 9 + pos % 7, ncap=2) * np.float32(0.3)
        x[1, pos] = bad
        c = _chain("mfma", 1e6, decim=D, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=2, max_samples=n)
        c.batch_process(x)
        y = c.fetch_demod().reshape(2, -1)
        c.close()
        want = odsp.fir_ccf(x[1], taps, D)
        wbad = ~np.isfinite(want)
        # outputs m with m*D - (ntaps-1) <= pos <= m*D
        m_lo, m_hi = -(-pos // D), min(want.size - 1, (pos + ntaps - 1) // D)
        assert wbad.sum() == max(0, m_hi - m_lo + 1) and (wbad.sum() == 0 or wbad[m_lo:m_hi + 1].all())
        assert np.array_equal(~np.isfinite(y[1]), wbad), (pos, int((~np.isfinite(y[1])).sum()), int(wbad.sum()))
        assert _same_or_both_nan(y[1], want)
        assert _same_words(y[0], odsp.fir_ccf(x[0], taps, D))


def test_mfma_nonfinite_through_the_discriminator_chain():
    """the same through k_frontend_mfma's demod + slicer epilogue: floats and sliced bytes == the oracle chain"""
    from sdr_amd import chain
    taps = odsp.low_pass(1.0, 2e6, 20e3, 40e3)      # 121 taps
    n = 3 * 1024 + 301
    x = _iq(n, 4, ncap=2) * np.float32(0.5)
    x[0, 1500] = np.inf
    x[1, 2000] = np.nan
    for dm, gain, lo, hi in ((chain.DEMOD_MAG2, 3.0, 0.05, 0.15), (chain.DEMOD_QUAD, 0.7, -0.3, 0.3)):
        c = _chain("mfma", 1e6, decim=1, taps=taps, demod=dm, demod_gain=gain, thr=(lo, hi), keep_n=1, tap_demod=True,
                   max_batch=2, max_samples=n)
        c.batch_process(x)
        v = c.fetch_demod().reshape(2, -1)
        b = c.fetch_bits().reshape(2, -1)
        c.close()
        for i in range(2):
            y = odsp.fir_ccf(x[i], taps, 1)
            wv = odsp.mag2(y) if dm == chain.DEMOD_MAG2 else odsp.quad_demod(y, 1.0)
            # outputs whose window (or, for the discriminator, whose predecessor's window) holds the sample
            touched = ~np.isfinite(odsp.mag2(y))
            touched[1:] |= touched[:-1].copy()
            assert 100 < touched.sum() < 130
            assert _same_or_both_nan(v[i][~touched], (np.float32(gain) * wv)[~touched])
            assert np.isfinite(v[i][~touched]).all()
            wb, _, _ = odsp.slice_bits(wv, gain, lo, hi, 1)
            if dm == chain.DEMOD_MAG2:
                assert _same_or_both_nan(v[i], np.float32(gain) * wv)
                np.testing.assert_array_equal(b[i], wb)
            else:
                # fast_atan2f of a NaN / Inf argument is whatever its comparisons make of it (the oracle's C branches
                # and the branchless device form need not agree there: parity unpinned, the reference holds no such
                # sample); the state the slicer is left in behind the stretch may therefore differ, so the bytes are
                # compared up to the first touched output only
                first = int(np.argmax(touched))
                np.testing.assert_array_equal(b[i][:first], wb[:first])
