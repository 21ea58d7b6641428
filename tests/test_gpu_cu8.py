"""rtl-sdr byte input: (u8 I, u8 Q) pairs converted while the
tile is staged, as rtlsdr_source does ((u8 - 127.4) / 128).  Parity bar: the
channel-filter floats are bit-identical to the oracle run on the converted
samples; sliced bytes and frames bit-exact."""
import numpy as np
import pytest

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import chain, segments, synth

pytestmark = pytest.mark.gpu

FS = 1e6


def _quantise(iq):
    f = np.ascontiguousarray(iq, np.complex64).view(np.float32)
    return np.clip(np.rint(f * 128.0 + 127.4), 0, 255).astype(np.uint8)


def test_cu8_conversion_is_exact_for_all_bytes():
    u = np.arange(256, dtype=np.uint8)
    z = odsp.cu8_to_fc32(np.stack([u, u[::-1]], 1))
    want = (u.astype(np.float64) - float(np.float32(127.4))) / 128.0
    assert np.array_equal(z.real.astype(np.float64), want)
    with chain.Chain(FS, demod=chain.DEMOD_NONE, tap_demod=True, max_samples=256, input_format="cu8") as c:
        c.batch_process(np.stack([u, u[::-1]], 1)[None])
        y = c.fetch_demod()
    assert np.array_equal(y.view(np.uint32), z.view(np.uint32))


@pytest.mark.parametrize("D,n,ncap", [(1, 4099, 1), (7, 30001, 3), (50, 1 << 16, 4), (50, 65530, 2), (200, 1 << 16, 2)])
def test_cu8_fir_float_parity(D, n, ncap):
    rng = np.random.default_rng(D + n)
    u = rng.integers(0, 256, (ncap, n, 2), dtype=np.uint8)
    taps = chain.low_pass(1.0, FS, 20e3, 40e3)
    with chain.Chain(FS, decim=D, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=ncap, max_samples=n,
                     input_format="cu8") as c:
        c.batch_process(u)
        y = c.fetch_demod().reshape(ncap, -1)
    for i in range(ncap):
        want = odsp.fir_ccf(odsp.cu8_to_fc32(u[i]), taps, D)
        assert np.array_equal(y[i].view(np.uint32), want.view(np.uint32)), i


def test_cu8_ts33c_chain_and_device_input():
    import torch
    n, D = 1 << 18, 50
    iq, pkt, _ = synth.ts33c_capture(n, FS, seed=5, f_off=100e3)
    u = _quantise(iq).reshape(n, 2)
    taps = chain.low_pass(1.0, FS, 20e3, 40e3)
    z = odsp.cu8_to_fc32(u)
    want = odsp.ook_chain(z, taps, D, 1.0, 0.05, 0.15, 1, fc=100e3, fs=FS)
    o = odec.TS33CStreamDecoder(sample_rate=20000)
    o.work([want])
    assert [f["packet"] for f in o.frames] == [pkt, pkt]
    kw = dict(decim=D, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_samples=n, input_format="cu8")
    with chain.Chain(FS, **kw) as c:
        c.batch_process(torch.from_numpy(u[None]).to("cuda:0"))
        assert np.array_equal(c.fetch_bits(), want)
        assert [f.packet() for f in c.fetch_frames()] == [pkt, pkt]
    # streaming, ragged chunks (history kept as bytes) == one shot
    with chain.Chain(FS, **kw) as c:
        rng = np.random.default_rng(1)
        cuts = np.sort(rng.integers(0, n, 9))
        got, frames = [], []
        for a, b in zip(np.r_[0, cuts], np.r_[cuts, n]):
            c.process_iq(u[a:b])
            got.append(c.fetch_bits())
            frames += c.fetch_frames()
        assert np.array_equal(np.concatenate(got), want)
        assert [f.packet() for f in frames] == [pkt, pkt]
    # segments with a byte halo
    with chain.Chain(FS, decim=D, center_freq=100e3, taps=taps, max_samples=n, input_format="cu8") as c:
        bits, infos = [], []
        for a, b in segments.segment_bounds(n, 4, 400):
            infos.append(segments.run_segment(c, u, a, b))
            bits.append(c.fetch_bits())
    assert np.array_equal(segments.stitch_segments(bits, infos, 0), want)


def test_cu8_prefilter_chain_takes_bytes():
    """mumbi FS300 cascade fed with bytes: the format applies to the upstream stage."""
    n = 1 << 16
    rng = np.random.default_rng(4)
    u = rng.integers(100, 156, (n, 2), dtype=np.uint8)
    lp = chain.low_pass(1.0, 2e6, 100e3, 50e3)
    rs = chain.resampler_taps(1, 10)
    with chain.Chain(2e6, decim=10, taps=rs, demod=chain.DEMOD_NONE, tap_demod=True, max_samples=n, pre_taps=lp,
                     pre_decim=1, input_format="cu8") as c:
        c.batch_process(u[None])
        y = c.fetch_demod()
    want = odsp.fir_ccf(odsp.fir_ccf(odsp.cu8_to_fc32(u), lp, 1), rs, 10)
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("fc", [0.0, 100e3])
@pytest.mark.parametrize("n,ncap,pad", [(30000 + 301, 3, 0), (30000 + 301, 3, 3), (8000, 2, 8), (100, 1, 0)])
def test_cu8_reference_wiring_on_the_window_kernel(fc, n, ncap, pad):
    """A recorded .cu8 file through the reference's own graph (TS33C/live_decoder.grc:917-988: 121 taps at
    decimation 1, keep 1-in-50 after the slicer), with and without a carrier offset: the sliding-window kernel
    converts the bytes while staging -- 16-B chunks on aligned rows (pad 0, 8: row strides of whole chunks,
    tiles that start anywhere inside a chunk), sample by sample otherwise (pad 3).  Floats and bytes against the
    oracle on the converted samples; streaming in ragged chunks on top."""
    from tests.test_gpu_mfma import _chain
    import ctypes as C
    from sdr_amd import _lib
    rng = np.random.default_rng(n + ncap + pad)
    st = ((n + 7) // 8) * 8 + pad
    flat = rng.integers(100, 156, (ncap * st, 2), dtype=np.uint8)
    taps = chain.low_pass(1.0, FS, 20e3, 40e3)
    res = {}
    for mode in ("win", "valu"):
        c = _chain(mode, FS, decim=1, center_freq=fc, taps=taps, demod=chain.DEMOD_MAG2, demod_gain=8.0, thr=(0.05, 0.15),
                   keep_n=50, tap_demod=True, max_batch=ncap, max_samples=n, input_format="cu8")
        _lib.check(c._h, c._L.sdr_batch_process(c._h, C.c_void_p(flat.ctypes.data), n, st, ncap, 0, C.byref(c.out)))
        res[mode] = (c.fetch_demod().reshape(ncap, -1), c.fetch_bits().reshape(ncap, -1))
        c.close()
    assert np.array_equal(res["win"][0].view(np.uint32), res["valu"][0].view(np.uint32))
    assert np.array_equal(res["win"][1], res["valu"][1])
    for i in range(ncap):
        z = odsp.cu8_to_fc32(flat[i * st:i * st + n])
        y = odsp.freq_xlating_fir(z, taps, fc, FS, 1) if fc else odsp.fir_ccf(z, taps, 1)
        v = odsp.mag2(y)
        assert np.array_equal(res["win"][0][i].view(np.uint32), (np.float32(8.0) * v).view(np.uint32))
        b, _, _ = odsp.slice_bits(v, 8.0, 0.05, 0.15, 50)
        assert np.array_equal(res["win"][1][i], b)
    # streaming: byte history carried between calls
    u = flat[:n]
    z = odsp.cu8_to_fc32(u)
    v = odsp.mag2(odsp.freq_xlating_fir(z, taps, fc, FS, 1) if fc else odsp.fir_ccf(z, taps, 1))
    c = _chain("win", FS, decim=1, center_freq=fc, taps=taps, demod=chain.DEMOD_MAG2, demod_gain=8.0, thr=(0.05, 0.15),
               keep_n=50, tap_demod=True, max_samples=n, input_format="cu8")
    cuts = sorted(set([0, 1, 2, 9, min(n, 2560), min(n, 2561), n // 2, n]))
    fl = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        c.process_iq(u[a:b])
        fl.append(c.fetch_demod())
    c.close()
    assert np.array_equal(np.concatenate(fl).view(np.uint32), (np.float32(8.0) * v).view(np.uint32))
