"""sdr_batch_process_static (the allocation-free entry point, include/sdrhip.h) against sdr_batch_process:
same sliced bytes, same frame records, same counts -- bit for bit -- on two calls in a row, the second with
a smaller input (the static array keeps the first call's contents).  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

from sdr_amd import _lib, chain, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _batch(ncap, n, fs, seed):
    rng = np.random.default_rng(seed)
    iq = np.zeros((ncap, n), np.complex64)
    for i in range(ncap):
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        sym = []
        for _ in range(2):
            sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
        env = synth.envelope(sym, fs, lead_in_us=5000.0 + 300.0 * rng.random(), total=n)
        iq[i] = synth.ook_iq(env, fs, f_off=100e3, amp=0.7, sigma=0.02, rng=rng)
    return iq


def _reference(c, x, max_frames):
    """the existing entry point + its device-side exports"""
    ncap = x.shape[0]
    c.batch_process(x)
    nb = int(c.out.n_bits)
    bits = torch.zeros(ncap * nb, dtype=torch.uint8, device=x.device)
    assert c.export_bits(bits) == ncap * nb
    rec = torch.zeros(ncap * max_frames * C.sizeof(_lib.Frame), dtype=torch.uint8, device=x.device)
    cnt = torch.zeros(ncap, dtype=torch.int32, device=x.device)
    c.export_frames(rec, cnt)
    c.synchronize()
    return nb, bits.cpu().numpy(), rec.cpu().numpy(), cnt.cpu().numpy()


def test_static_entry_point_matches_batch_process_twice():
    fs, D, max_frames = 1e6, 50, 4
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    dev = torch.device("cuda", 0)
    L = _lib.load()
    assert L.sdr_static_scratch_bytes() == 8 << 20
    calls = [(24, 1 << 18, 5), (7, 180000, 6)]   # second call: fewer and shorter captures
    with chain.Chain(fs, decim=D, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_batch=32,
                     max_frames=max_frames, max_samples=1 << 18) as ref, \
            chain.Chain(fs, decim=D, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_batch=32,
                        max_frames=max_frames, max_samples=1 << 18) as c:
        for ncap, n, seed in calls:
            x = torch.from_numpy(_batch(ncap, n, fs, seed).view(np.float32).reshape(ncap, n, 2)).to(dev)
            nb, want_bits, want_rec, want_cnt = _reference(ref, x, max_frames)
            # outputs pre-filled with a pattern: everything the call owns must be overwritten
            bits = torch.full((ncap * nb,), 0xAB, dtype=torch.uint8, device=dev)
            rec = torch.full((ncap * max_frames * C.sizeof(_lib.Frame),), 0xCD, dtype=torch.uint8, device=dev)
            cnt = torch.full((ncap,), -7, dtype=torch.int32, device=dev)
            out = c.batch_process_static(x, bits, rec, cnt)
            c.synchronize()
            assert int(out.n_bits) == nb and int(out.n_in) == ncap * n
            assert np.array_equal(cnt.cpu().numpy(), want_cnt) and want_cnt.sum() >= ncap
            assert np.array_equal(bits.cpu().numpy(), want_bits)
            assert np.array_equal(rec.cpu().numpy(), want_rec)
        # sizes beyond the static array are refused with a clear error, nothing is launched
        big = torch.zeros((32, 1 << 18, 2), dtype=torch.float32, device=dev)
        with chain.Chain(fs, decim=1, taps=taps, decoder=chain.DEC_TS33C, max_batch=32, max_frames=max_frames,
                         max_samples=1 << 18) as wide:
            bits = torch.zeros(32 << 18, dtype=torch.uint8, device=dev)
            rec = torch.zeros(32 * max_frames * C.sizeof(_lib.Frame), dtype=torch.uint8, device=dev)
            cnt = torch.zeros(32, dtype=torch.int32, device=dev)
            with pytest.raises(_lib.SdrError) as ei:
                wide.batch_process_static(big, bits, rec, cnt)
            assert ei.value.code == _lib.SDR_E_CAPACITY and "static array" in str(ei.value)


def test_static_entry_point_under_graph_capture():
    """One of the callers the entry point exists for: captured into a HIP graph (torch.cuda.graph), nothing runs at
    capture time; every replay gives the bytes, records and counts sdr_batch_process gives for the input tensor's
    contents at that moment."""
    fs, D, max_frames = 1e6, 50, 4
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    dev = torch.device("cuda", 0)
    ncap, n = 24, 1 << 18
    kw = dict(decim=D, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_batch=32, max_frames=max_frames, max_samples=n)
    with chain.Chain(fs, **kw) as ref, chain.Chain(fs, **kw) as c:
        batches = [torch.from_numpy(_batch(ncap, n, fs, seed).view(np.float32).reshape(ncap, n, 2)).to(dev) for seed in (31, 32)]
        x = batches[0].clone()
        nb = n // D
        bits = torch.zeros(ncap * nb, dtype=torch.uint8, device=dev)
        rec = torch.zeros(ncap * max_frames * C.sizeof(_lib.Frame), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(ncap, dtype=torch.int32, device=dev)
        st = torch.cuda.Stream(device=dev)
        c.set_stream(st.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            c.batch_process_static(x, bits, rec, cnt)
        torch.cuda.synchronize()
        assert int(cnt.sum()) == 0 and int(bits.sum()) == 0          # captured, not run
        for b in batches + batches[:1]:
            x.copy_(b)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            want_nb, want_bits, want_rec, want_cnt = _reference(ref, b, max_frames)
            assert want_nb == nb and want_cnt.sum() >= ncap
            assert np.array_equal(cnt.cpu().numpy(), want_cnt)
            assert np.array_equal(bits.cpu().numpy(), want_bits)
            assert np.array_equal(rec.cpu().numpy(), want_rec)
        del g


def test_static_entry_point_two_handles_two_streams_back_to_back():
    """Every call on a GPU shares that GPU's static array.  Two handles on two different streams, called back to
    back with nothing between them on the host: the second call is queued behind the first one's event on the
    device (round 3), so both give what sdr_batch_process gives -- before that they raced."""
    fs, D, max_frames = 1e6, 50, 4
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    dev = torch.device("cuda", 0)
    ncap, n = 28, 1 << 18
    kw = dict(decim=D, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_batch=32, max_frames=max_frames, max_samples=n)
    xs = [torch.from_numpy(_batch(ncap, n, fs, seed).view(np.float32).reshape(ncap, n, 2)).to(dev) for seed in (41, 42)]
    nb = n // D
    with chain.Chain(fs, **kw) as ref, chain.Chain(fs, **kw) as a, chain.Chain(fs, **kw) as b:
        want = [_reference(ref, x, max_frames) for x in xs]
        outs = [(torch.zeros(ncap * nb, dtype=torch.uint8, device=dev),
                 torch.zeros(ncap * max_frames * C.sizeof(_lib.Frame), dtype=torch.uint8, device=dev),
                 torch.zeros(ncap, dtype=torch.int32, device=dev)) for _ in xs]
        torch.cuda.synchronize()
        for rep in range(6):                       # each handle has its own stream (sdr_create)
            for c, x, o in ((a, xs[0], outs[0]), (b, xs[1], outs[1])):
                c.batch_process_static(x, *o)
        a.synchronize()
        b.synchronize()
        for (wnb, wbits, wrec, wcnt), (bits, rec, cnt) in zip(want, outs):
            assert wnb == nb and wcnt.sum() >= ncap
            assert np.array_equal(cnt.cpu().numpy(), wcnt)
            assert np.array_equal(bits.cpu().numpy(), wbits)
            assert np.array_equal(rec.cpu().numpy(), wrec)


def test_zz_guard_bands_behind_device_buffers():
    """Collected last.  Under SDR_POISON=1 every device buffer the library allocated had a 512-byte guard band
    behind it, checked when the buffer was freed: none may have been written.  (Without the switch: nothing to
    check, the counter stays 0.)"""
    import gc
    import os
    gc.collect()                      # handles of earlier tests are freed by now
    L = _lib.load()
    L.sdr_selftest_guard_hits.restype = C.c_int
    L.sdr_selftest_guard_hits.argtypes = [C.POINTER(C.c_int)]
    checked = C.c_int(0)
    assert L.sdr_selftest_guard_hits(C.byref(checked)) == 0
    if os.environ.get("SDR_POISON"):
        assert checked.value > 100
