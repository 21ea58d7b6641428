"""Multi-GPU plumbing: captures are independent, so the only collective is the
gather of decoded frame records and per-capture counts.  Works on
any torch.distributed backend: "nccl" (= RCCL over xGMI) in bench.py, "gloo" in
the CPU tests."""
import ctypes as C

import numpy as np

from . import _lib

FRAME_BYTES = C.sizeof(_lib.Frame)


def shard_range(n_total, rank, world):
    """Contiguous block of capture indices owned by `rank` (i -> i // ceil)."""
    per = (n_total + world - 1) // world
    a = min(n_total, rank * per)
    return a, min(n_total, a + per)


def gather_frames(rec, cnt, world, all_rec=None, all_cnt=None):
    """all_gather of fixed-size frame tables.  rec: uint8 [ncap*max_frames*96],
    cnt: int32 [ncap] on this rank's device.  Returns (all_rec, all_cnt)."""
    import torch
    import torch.distributed as dist
    if all_rec is None:
        all_rec = torch.zeros(world * rec.numel(), dtype=rec.dtype, device=rec.device)
    if all_cnt is None:
        all_cnt = torch.zeros(world * cnt.numel(), dtype=cnt.dtype, device=cnt.device)
    dist.all_gather_into_tensor(all_cnt, cnt)
    dist.all_gather_into_tensor(all_rec, rec)
    return all_rec, all_cnt


def unpack_frames(all_rec, all_cnt, ncap_per_rank, max_frames):
    """Host side of the gather: -> list of (rank, Frame) in rank/capture order."""
    rec = np.ascontiguousarray(all_rec.cpu().numpy())
    cnt = all_cnt.cpu().numpy()
    out = []
    for g in range(cnt.size):
        rank, cap = divmod(g, ncap_per_rank)
        for k in range(int(cnt[g])):
            off = ((rank * ncap_per_rank + cap) * max_frames + k) * FRAME_BYTES
            out.append((rank, _lib.Frame.from_buffer_copy(rec[off:off + FRAME_BYTES].tobytes())))
    return out
