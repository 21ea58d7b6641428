"""N>1 path on CPU: world_size-2 gloo run of the frame gather that bench.py does
over RCCL (fixed-size frame tables + per-capture counts), and the shard map."""
import ctypes as C
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdr_amd import _lib, dist as sdist


def test_shard_range_covers_everything():
    for n, w in [(10000, 8), (10, 3), (5, 8), (0, 2)]:
        seen = []
        for r in range(w):
            a, b = sdist.shard_range(n, r, w)
            seen += list(range(a, b))
        assert seen == list(range(n))
    assert sdist.shard_range(10000, 3, 8) == (3750, 5000)


def _fake_table(rank, ncap, max_frames):
    rec = np.zeros((ncap, max_frames, sdist.FRAME_BYTES), np.uint8)
    cnt = np.zeros(ncap, np.int32)
    for c in range(ncap):
        cnt[c] = (c + rank) % (max_frames + 1)
        for k in range(cnt[c]):
            f = _lib.Frame()
            f.capture, f.decoder, f.status, f.nbytes = c, 1, 0, 10
            for j in range(10):
                f.bytes[j] = (rank * 97 + c * 7 + k * 3 + j) & 0xff
            rec[c, k] = np.frombuffer(bytes(f), np.uint8)
    return rec, cnt


def _worker(rank, world, port, ncap, max_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec, cnt = _fake_table(rank, ncap, max_frames)
    all_rec, all_cnt = sdist.gather_frames(torch.from_numpy(rec.reshape(-1)), torch.from_numpy(cnt), world)
    frames = sdist.unpack_frames(all_rec, all_cnt, ncap, max_frames)
    q.put((rank, [(r, f.capture, bytes(bytearray(f.bytes[:10]))) for r, f in frames]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_gather_world2():
    world, ncap, max_frames = 2, 5, 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ncap, max_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = []
    for r in range(world):
        rec, cnt = _fake_table(r, ncap, max_frames)
        for c in range(ncap):
            for k in range(cnt[c]):
                f = _lib.Frame.from_buffer_copy(rec[c, k].tobytes())
                want.append((r, f.capture, bytes(bytearray(f.bytes[:10]))))
    assert got[0] == got[1] == want and len(want) > 5
