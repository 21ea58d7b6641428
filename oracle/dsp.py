"""ctypes front end of ``liboracle_dsp.so`` (oracle/dsp_oracle.c) -- TEST INFRASTRUCTURE.

Each wrapper takes/returns numpy arrays and handles the history padding the C
functions expect (``ntaps-1`` zeros in front of a stream start).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WIN_HAMMING, WIN_HANN, WIN_BLACKMAN, WIN_RECTANGULAR, WIN_KAISER, WIN_BLACKMAN_HARRIS = range(6)

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_dsp.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle library not built: run `make -C oracle` "
                               "(or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_firdes_low_pass.restype = C.c_int
        L.orc_firdes_low_pass.argtypes = [C.c_double] * 4 + [C.c_int, C.c_double, _f32p, C.c_int]
        L.orc_resampler_design.restype = C.c_int
        L.orc_resampler_design.argtypes = [C.c_int, C.c_int, C.c_double, _f32p, C.c_int]
        for name in ("orc_fir_ccf", "orc_fir_fff", "orc_fir_ccc"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_int, C.c_int, C.c_int64, _f32p]
        L.orc_xlating_taps.restype = None
        L.orc_xlating_taps.argtypes = [_f32p, C.c_int, C.c_double, C.c_double, _f32p]
        L.orc_rotator_gr.restype = None
        L.orc_rotator_gr.argtypes = [_f32p, C.c_int64, C.c_double, C.c_double, C.c_int, _f32p,
                                     C.POINTER(C.c_int64)]
        L.orc_nco_inc.restype = C.c_uint64
        L.orc_nco_inc.argtypes = [C.c_double, C.c_double, C.c_int]
        L.orc_nco_table.restype = None
        L.orc_nco_table.argtypes = [_f32p]
        L.orc_rotate_nco.restype = None
        L.orc_rotate_nco.argtypes = [_f32p, C.c_int64, C.c_uint64, C.c_uint64, _f32p]
        L.orc_mag2.restype = None
        L.orc_mag2.argtypes = [_f32p, C.c_int64, _f32p]
        L.orc_atan_table.restype = None
        L.orc_atan_table.argtypes = [_f32p]
        L.orc_fast_atan2f.restype = C.c_float
        L.orc_fast_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_quad_demod.restype = None
        L.orc_quad_demod.argtypes = [C.c_void_p, C.c_int64, C.c_float, _f32p]
        L.orc_fm_deemph_coeffs.restype = None
        L.orc_fm_deemph_coeffs.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double)]
        L.orc_iir_deemph.restype = None
        L.orc_iir_deemph.argtypes = [_f32p, C.c_int64, C.c_double, C.c_double,
                                     np.ctypeslib.ndpointer(np.float64), _f32p]
        L.orc_pwr_squelch.restype = C.c_int64
        L.orc_pwr_squelch.argtypes = [_f32p, C.c_int64, C.c_double, C.c_double, C.c_int, _f64p, _f32p]
        L.orc_slice.restype = C.c_int64
        L.orc_slice.argtypes = [_f32p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                                C.POINTER(C.c_int), C.POINTER(C.c_int), _i8p]
        L.orc_resample_ccc.restype = C.c_int64
        L.orc_resample_ccc.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_int, C.c_int, C.c_int,
                                       _f32p, C.c_int]
        L.orc_ook_chain.restype = C.c_int64
        L.orc_ook_chain.argtypes = [_f32p, C.c_int64, _f32p, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                    C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, _i8p,
                                    C.c_void_p]
        L.orc_pfb_twiddles.restype = None
        L.orc_pfb_twiddles.argtypes = [C.c_int, _f32p]
        L.orc_pfb_channelize.restype = C.c_int
        L.orc_pfb_channelize.argtypes = [_f32p, C.c_int64, _f32p, C.c_int, C.c_int, _f32p, C.c_int64, _f32p]
        _LIB = L
    return _LIB


def _c2f(x):
    """complex64 array -> interleaved float32 view (copy if needed)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    return x.view(np.float32)


def low_pass(gain, fs, fc, tw, win=WIN_HAMMING, beta=6.76):
    buf = np.zeros(1 << 16, np.float32)
    n = lib().orc_firdes_low_pass(gain, fs, fc, tw, win, beta, buf, buf.size)
    assert n > 0
    return buf[:n].copy()


def resampler_taps(interp, decim, fbw=0.4):
    g = np.gcd(interp, decim)
    buf = np.zeros(1 << 16, np.float32)
    n = lib().orc_resampler_design(interp // g, decim // g, fbw, buf, buf.size)
    assert n > 0
    return buf[:n].copy()


def xlating_taps(taps, fc, fs):
    taps = np.ascontiguousarray(taps, np.float32)
    out = np.zeros(2 * taps.size, np.float32)
    lib().orc_xlating_taps(taps, taps.size, fc, fs, out)
    return out.view(np.complex64)


def nco_table():
    t = np.zeros(2048, np.float32)
    lib().orc_nco_table(t)
    return t


def nco_inc(fc, fs, D=1):
    return int(lib().orc_nco_inc(fc, fs, D))


def _with_hist(xf, H, hist, width):
    buf = np.zeros(xf.size + H * width, np.float32)
    if hist is not None and H:
        buf[:H * width] = np.ascontiguousarray(hist).view(np.float32)[-H * width:]
    buf[H * width:] = xf
    return buf


def fir_ccf(x, taps, D=1, hist=None, phase0=0, nout=None):
    taps = np.ascontiguousarray(taps, np.float32)
    xf = _c2f(x)
    H = taps.size - 1
    buf = _with_hist(xf, H, hist, 2)
    n = x.size
    if nout is None:
        nout = (n - phase0) // D
    y = np.zeros(2 * nout, np.float32)
    lib().orc_fir_ccf(buf.ctypes.data + 8 * H, nout, taps, taps.size, D, phase0, y)
    return y.view(np.complex64)


def fir_ccc(x, ctaps, D=1, hist=None, phase0=0, nout=None):
    ct = _c2f(ctaps)
    ntaps = ct.size // 2
    xf = _c2f(x)
    H = ntaps - 1
    buf = _with_hist(xf, H, hist, 2)
    if nout is None:
        nout = (x.size - phase0) // D
    y = np.zeros(2 * nout, np.float32)
    lib().orc_fir_ccc(buf.ctypes.data + 8 * H, nout, ct, ntaps, D, phase0, y)
    return y.view(np.complex64)


def fir_fff(x, taps, D=1, hist=None, phase0=0, nout=None):
    taps = np.ascontiguousarray(taps, np.float32)
    xf = np.ascontiguousarray(x, np.float32)
    H = taps.size - 1
    buf = _with_hist(xf, H, hist, 1)
    if nout is None:
        nout = (xf.size - phase0) // D
    y = np.zeros(nout, np.float32)
    lib().orc_fir_fff(buf.ctypes.data + 4 * H, nout, taps, taps.size, D, phase0, y)
    return y


def rotate_nco(y, inc, m0=0):
    yf = _c2f(y).copy()
    lib().orc_rotate_nco(yf, yf.size // 2, inc, m0, nco_table())
    return yf.view(np.complex64)


def rotate_gr(y, fc, fs, D=1):
    yf = _c2f(y).copy()
    st = np.array([1.0, 0.0], np.float32)
    ctr = C.c_int64(0)
    lib().orc_rotator_gr(yf, yf.size // 2, fc, fs, D, st, C.byref(ctr))
    return yf.view(np.complex64)


def freq_xlating_fir(x, taps, fc, fs, D=1, rotator="nco64"):
    """freq_xlating_fir_filter_ccc: composite FIR then rotate (a1)."""
    taps = np.atleast_1d(np.asarray(taps, np.float32))
    ct = xlating_taps(taps, fc, fs)
    y = fir_ccc(x, ct, D)
    if rotator == "nco64":
        return rotate_nco(y, nco_inc(fc, fs, D))
    return rotate_gr(y, fc, fs, D)


def mag2(x):
    xf = _c2f(x)
    y = np.zeros(xf.size // 2, np.float32)
    lib().orc_mag2(xf, y.size, y)
    return y


def atan_table():
    t = np.zeros(257, np.float32)
    lib().orc_atan_table(t)
    return t


def fast_atan2f(y, x):
    return float(lib().orc_fast_atan2f(y, x))


def quad_demod(x, gain, prev=0j):
    xf = _c2f(x)
    buf = np.zeros(xf.size + 2, np.float32)
    buf[0], buf[1] = np.float32(prev.real), np.float32(prev.imag)
    buf[2:] = xf
    y = np.zeros(xf.size // 2, np.float32)
    lib().orc_quad_demod(buf.ctypes.data + 8, y.size, gain, y)
    return y


def deemph_coeffs(fs, tau=75e-6):
    b0, p1 = C.c_double(), C.c_double()
    lib().orc_fm_deemph_coeffs(fs, tau, C.byref(b0), C.byref(p1))
    return b0.value, p1.value


def fm_deemph(x, fs, tau=75e-6, state=None):
    b0, p1 = deemph_coeffs(fs, tau)
    st = np.zeros(2, np.float64) if state is None else state
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    lib().orc_iir_deemph(x, x.size, b0, p1, st, y)
    return y


def pwr_squelch(x, db, alpha, gate=True, state=None):
    """analog_pwr_squelch_cc, ramp 0 (a7).  state: float64[1] power carried between calls."""
    xf = _c2f(x)
    y = np.zeros_like(xf)
    st = np.zeros(1, np.float64) if state is None else state
    k = lib().orc_pwr_squelch(xf, xf.size // 2, db, alpha, int(gate), st, y)
    return y[:2 * k].view(np.complex64)


def slice_bits(v, gain, lo, hi, keep_n=1, state=0, counter=None):
    """multiply_const->threshold_ff->add_const->binary_slicer->keep_one_in_n (a8)."""
    v = np.ascontiguousarray(v, np.float32)
    st = C.c_int(state)
    ctr = C.c_int(keep_n if counter is None else counter)
    out = np.zeros(v.size // max(1, keep_n) + 2, np.int8)
    k = lib().orc_slice(v, v.size, gain, lo, hi, keep_n, C.byref(st), C.byref(ctr), out)
    return out[:k].copy(), st.value, ctr.value


def resample(x, taps, L, M, hist=None):
    taps = np.ascontiguousarray(taps, np.float32)
    is_c = np.iscomplexobj(x)
    nt = (taps.size + L - 1) // L
    H = nt - 1
    if is_c:
        xf = _c2f(x)
        buf = _with_hist(xf, H, hist, 2)
        n = xf.size // 2
        y = np.zeros(2 * (n * L // M + 2), np.float32)
        k = lib().orc_resample_ccc(buf.ctypes.data + 8 * H, n, taps, taps.size, L, M, y, 1)
        return y[:2 * k].view(np.complex64).copy()
    xf = np.ascontiguousarray(x, np.float32)
    buf = _with_hist(xf, H, hist, 1)
    y = np.zeros(xf.size * L // M + 2, np.float32)
    k = lib().orc_resample_ccc(buf.ctypes.data + 4 * H, xf.size, taps, taps.size, L, M, y, 0)
    return y[:k].copy()


def pfb_channelize(x, taps, M=64):
    """Critically sampled polyphase channelizer from stream start ->
    complex64 [M, len(x)//M]; row c = the output of
    freq_xlating_fir(x, taps, c*fs/M, fs, D=M) (see orc_pfb_channelize)."""
    xf = _c2f(x)
    n = xf.size // 2
    taps = np.ascontiguousarray(taps, np.float32)
    tw = np.zeros(2 * M, np.float32)
    lib().orc_pfb_twiddles(M, tw)
    nout = n // M
    y = np.zeros(2 * M * max(nout, 1), np.float32)
    rc = lib().orc_pfb_channelize(xf, n, taps, taps.size, M, tw, nout, y)
    assert rc == 0
    return y[:2 * M * nout].view(np.complex64).reshape(M, nout)


def cu8_to_fc32(u8):
    """rtlsdr_source's conversion of the dongle's bytes [GR-ext: gr-osmosdr
    lib/rtl/rtl_source_c.cc lookup table, unpinned]: (float(u8) - 127.4f) * (1/128)
    per component.  u8: (..., 2) or flat interleaved I,Q -> complex64."""
    f = (np.asarray(u8, np.uint8).astype(np.float32) - np.float32(127.4)) * np.float32(1.0 / 128.0)
    return np.ascontiguousarray(f.reshape(-1, 2)).view(np.complex64).reshape(-1)


def ook_chain(iq, taps, D, gain, lo, hi, keep_n, fc=0.0, fs=1.0, return_mag=False):
    """Whole OOK chain from stream start; taps real (fc==0) or translated (fc!=0)."""
    xf = _c2f(iq)
    n = xf.size // 2
    taps = np.atleast_1d(np.asarray(taps, np.float32))
    if fc != 0.0:
        ct = _c2f(xlating_taps(taps, fc, fs))
        tab = nco_table()
        tabp = tab.ctypes.data
        inc = nco_inc(fc, fs, D)
        tc = 1
    else:
        ct, tabp, inc, tc = taps, None, 0, 0
    nout = n // D
    bits = np.zeros(nout // keep_n + 2, np.int8)
    mag = np.zeros(max(nout, 1), np.float32) if return_mag else None
    k = lib().orc_ook_chain(xf, n, ct, taps.size, tc, D, inc, tabp, gain, lo, hi, keep_n, bits,
                            mag.ctypes.data if return_mag else None)
    if return_mag:
        return bits[:k].copy(), mag[:nout]
    return bits[:k].copy()


def fm_radio_chain(iq, samp_rate=2e6, shift=200e3, volume=1.0, lpf=(2e6, 400e3, 200e3), quad_rate=500e3,
                   audio_decimation=10, tau=75e-6, audio_resample=(48, 50), rotator="nco64"):
    """FM-Radio/FM-Radio.grc:2903-2964 block after block (each its own pass, as
    the GNU Radio scheduler runs it).  Returns the five stage outputs."""
    D = int(samp_rate / 500e3)
    x = freq_xlating_fir(iq, [1.0], -shift, samp_rate, 1, rotator=rotator)   # "gr": GNU Radio's float recurrence
    y0 = fir_ccf(x, resampler_taps(1, D) if D > 1 else np.ones(1, np.float32), D)
    y1 = fir_ccf(y0, low_pass(1.0, lpf[0], lpf[1], lpf[2], WIN_BLACKMAN), 1)
    gain = np.float32(quad_rate / (2 * np.pi * 75e3))
    q = quad_demod(y1, gain)
    audio_rate = quad_rate / audio_decimation
    width = audio_rate / 32
    a2 = fir_fff(q, low_pass(1.0, quad_rate, audio_rate / 2 - width, width, WIN_HAMMING), audio_decimation)
    a3 = fm_deemph(a2, audio_rate, tau)
    g = int(np.gcd(audio_resample[0], audio_resample[1]))
    L, M = audio_resample[0] // g, audio_resample[1] // g
    a4 = resample(a3, resampler_taps(L, M), L, M) * np.float32(volume)
    return y0, q, a2, a3, a4.astype(np.float32)
