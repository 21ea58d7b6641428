"""N>1 long-capture path on CPU (gloo, world size 2 and 3): the cross-segment
hysteresis scan + gather of sdr_amd.segments.  The per-segment front end is
played by the oracle here (no GPU): each rank slices ITS segment of the
demodulated stream with carry-in 0 and reports what libsdrhip's
sdr_segment_info_get reports; the gathered stream must equal the oracle's single
pass and decode to the same frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import segments, synth

FS, D, KEEP = 1e6, 10, 5


def _capture():
    n = 1 << 18
    iq, pkt, _ = synth.ts33c_capture(n, FS, seed=12)
    # a stretch that sits between the thresholds across the world-2 boundary,
    # entered from a pulse: the second segment starts undetermined
    k = n // 2
    iq[k - 3000:k - 1500] = 0.7
    iq[k - 1500:k + 2500] = 0.3
    return iq, pkt


def _segment_model(mag, a_out, b_out, lo, hi, keep):
    """What one rank's handle yields for outputs [a_out, b_out): bytes computed
    with carry-in 0 plus the summary of sdr_segment_info_get."""
    v = mag[a_out:b_out]
    evt = np.where(v > hi, 1, np.where(v < lo, 2, 0))
    idx = np.nonzero(evt)[0]
    st, state = np.zeros(v.size, np.int8), 0
    for i in range(v.size):
        if evt[i] == 1:
            state = 1
        elif evt[i] == 2:
            state = 0
        st[i] = state
    emit = [o for o in range(a_out, b_out) if (o + 1) % keep == 0]
    bits = np.array([st[o - a_out] for o in emit], np.int8)
    fe = int(idx[0]) + a_out if idx.size else -1
    und = len(emit) if fe < 0 else sum(1 for o in emit if o < fe)
    return bits, und, int(idx.size > 0), int(state)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    iq, _ = _capture()
    taps = odsp.low_pass(1.0, FS, 20e3, 40e3)
    _, mag = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, KEEP, return_mag=True)
    a, b = segments.segment_bounds(iq.size, world, D * KEEP)[rank]
    bits, und, he, ls = _segment_model(mag, a // D, b // D, np.float32(0.05), np.float32(0.15), KEEP)
    stream, offs = segments.gather_stream(torch.from_numpy(bits.copy()), bits.size, und, he, ls, world, 0)
    q.put((rank, stream.numpy().copy(), offs, und))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_long_capture_stitch(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, stream, offs, und = q.get(timeout=180)
        got[r] = (stream, offs, und)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    iq, pkt = _capture()
    taps = odsp.low_pass(1.0, FS, 20e3, 40e3)
    want = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, KEEP)
    for r in range(world):
        assert np.array_equal(got[r][0], want), r
    if world == 2:
        assert got[1][2] > 10  # the boundary really was undetermined
    o1 = odec.TS33CStreamDecoder(sample_rate=FS / D / KEEP)
    o1.work([want])
    o2 = odec.TS33CStreamDecoder(sample_rate=FS / D / KEEP)
    o2.work([got[0][0]])
    assert [f["packet"] for f in o1.frames] == [f["packet"] for f in o2.frames]


def test_resolve_carries_and_bounds():
    assert segments.resolve_carries([1, 0, 0, 1, 0], [1, 0, 0, 0, 1], 0) == [0, 1, 1, 1, 0]
    assert segments.resolve_carries([0, 0], [0, 0], 1) == [1, 1]
    for n, w, al in [(1000, 8, 50), (1 << 20, 8, 50), (10, 4, 50), (0, 2, 1)]:
        b = segments.segment_bounds(n, w, al)
        assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
        for (a0, a1), (b0, b1) in zip(b, b[1:]):
            assert a1 == b0 and (a1 % al == 0 or a1 == n)


# ---- configs[4]: the 64 channel rows of a wideband capture channelized in segments (segments.gather_rows)
PFB_FS, PFB_M, PFB_KEEP, PFB_N = 20e6, 64, 4, 1 << 18


def _pfb_mags():
    """|y_c[m]|^2 of the oracle channelizer over the whole capture (every rank can afford it at this size); the
    per-segment front end is modelled from it exactly as in _segment_model."""
    rng = np.random.default_rng(5)
    sym = []
    for _ in range(6):
        sym += [(1, float(rng.integers(300, 1500))), (0, float(rng.integers(300, 1500)))]
    em = [(7 * PFB_FS / PFB_M, sym, 500.0), ((40 - 64) * PFB_FS / PFB_M, sym[::-1], 2000.0), (21 * PFB_FS / PFB_M, sym * 2, 100.0)]
    x = synth.wideband_capture(PFB_N, PFB_FS, em, seed=6)
    taps = odsp.low_pass(1.0, PFB_FS, 100e3, 100e3)
    y = odsp.pfb_channelize(x, taps, PFB_M)
    return np.stack([odsp.mag2(y[c]) for c in range(PFB_M)])


def _pfb_worker(rank, world, port, lo, hi, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mags = _pfb_mags()
    a, b = segments.segment_bounds(PFB_N, world, PFB_M * PFB_KEEP)[rank]
    rows, und, he, ls = [], [], [], []
    for c in range(PFB_M):
        bits, u, h, l = _segment_model(mags[c], a // PFB_M, b // PFB_M, np.float32(lo), np.float32(hi), PFB_KEEP)
        rows.append(bits)
        und.append(u)
        he.append(h)
        ls.append(l)
    bits = torch.from_numpy(np.stack(rows).copy())
    stream, offs = segments.gather_rows(bits, bits.shape[1], und, he, ls, world, 0)
    q.put((rank, stream.numpy().copy(), offs, int(sum(und))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lo,hi", [(0.05, 0.15), (1e-12, 0.05)])
def test_channelizer_rows_stitch_world2(lo, hi):
    """64 rows, world size 2: per-row carries resolved across the segment boundary; the second threshold pair
    leaves idle channels between the thresholds, so their second segment starts undetermined"""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pfb_worker, args=(r, world, port, lo, hi, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, stream, offs, und = q.get(timeout=240)
        got[r] = (stream, offs, und)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mags = _pfb_mags()
    want = np.stack([odsp.slice_bits(mags[c], 1.0, lo, hi, PFB_KEEP)[0] for c in range(PFB_M)])
    for r in range(world):
        assert got[r][0].shape == want.shape and np.array_equal(got[r][0], want), r
    assert want[7].any() and want[21].any()
    if lo < 1e-6:
        assert got[1][2] > 100   # rows of the second segment really started undetermined
