"""Frames-only exchange for one long capture cut into segments (sdr_amd.segments: stitch_tables / keep_local /
gather_frames; north_star "RCCL ... used only to gather decoded frames and counts") -- on CPU.

The per-segment pass of libsdrhip (sdr_segment_decode) is played here by a numpy model of the same rule on the
oracle's bytes plus the oracle's decoder; the joined table is turned back into bytes and decoded by the oracle, as
sdr_process_runs does on the GPU.  Frames (packet, status, limits, end position) must equal the oracle's single
work() call on the whole stream (TS33C/stream_decoder.py:39-220, TFA/stream_decoder.py:59-218), for random cuts,
cuts inside packets, inside held pulses and inside the gaps between packets."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import decoders as odec
from sdr_amd import _lib, segments, synth

TEXT_STATUS = {v: k for k, v in _lib.STATUS_TEXT.items()}
TEXT_STATUS["exc"] = _lib.FRAME_EXC_RESHAPE


class _TFA(odec.TFAStreamDecoder):
    """A packet cut by a segment boundary can run into the reference's latent exceptions (odd sync count:
    TFA/stream_decoder.py:137, no data start: :143) in the segment's OWN pass; libsdrhip records those as frames
    with an exception status and goes on, and so does this model.  Such frames belong to a head or a tail and
    are dropped by the keep window."""

    def _decode_packet(self):
        try:
            odec.TFAStreamDecoder._decode_packet(self)
        except (ValueError, IndexError):
            self.frames.append({"status": "exc", "packet": None, "symbol_length": 0.0})


def pieces_model(bits, rate, limit, cap):
    """sdr_segment_decode's head / tail rule (sdr_amd/csrc/sdrhip.hip) restated in numpy."""
    nb = int(bits.size)
    if nb == 0:
        return segments.LocalPieces(0, [], [], 0, 0, 0, 0, 0)
    b = np.asarray(bits, np.int8)
    pos = np.concatenate([[0], np.nonzero(np.diff(b))[0] + 1, [nb]]).astype(np.int64)
    R = pos.size - 1
    over = np.diff(pos) * 1e6 / rate > limit
    fl, ll = int(b[0] != 0), int(b[-1] != 0)
    nh = min(R + 1, cap)
    idx = np.nonzero(over[:nh - 1])[0]
    i1 = int(idx[0]) if idx.size else -1
    if i1 < 0 or i1 == R - 1:
        if nh < R + 1:
            return segments.LocalPieces(nb, [], [], fl, ll, 0, 0, 1)
        return segments.LocalPieces(nb, pos, [], fl, ll, 0, 0, 0)
    head = pos[:i1 + 2]
    span = R - i1
    nt = min(span, cap)
    lo = R + 1 - nt                       # tail window = pos[lo .. R]
    cand = [k for k in range(R - 2, lo - 1, -1) if over[k]]
    if cand:
        j = cand[0]
    elif nt < span:
        return segments.LocalPieces(nb, [], [], fl, ll, 0, 0, 2)
    else:
        j = i1
    tail = pos[j + 1:]
    return segments.LocalPieces(nb, head, tail, fl, ll, int(head[-1]), int(tail[0]), 0)


def to_frame(of):
    f = _lib.Frame()
    f.status = TEXT_STATUS[of["status"]]
    f.nbits = int(of.get("nbits", 0))
    pk = of["packet"] or b""
    f.nbytes = len(pk)
    for i, v in enumerate(pk[:16]):
        f.bytes[i] = v
    f.lim[0] = float(of.get("pulse_short_limit", of.get("symbol_length", 0.0)))
    f.lim[1] = float(of.get("gap_short_limit", 0.0))
    f.end_pos = of["end_pos"]
    return f


def key(f):
    return (int(f.end_pos), int(f.status), f.packet(), int(f.nbits), f.lim[0], f.lim[1])


def decode_table(make, pos, first_level):
    """ONE work() call of a fresh decoder on the stream the run table describes"""
    if pos.size < 2:
        return []
    w = np.diff(pos)
    lv = (first_level + np.arange(w.size)) & 1
    o = make()
    o.work([np.repeat(lv, w).astype(np.int8)])
    return [to_frame(f) for f in o.frames]


def frames_by_segments(make, bits, cuts, rate, limit, cap=1024):
    pl, local = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = bits[a:b]
        pl.append(pieces_model(seg, rate, limit, cap))
        o = make()
        if seg.size:
            o.work([seg])
        local.append([to_frame(f) for f in o.frames])
    if any(p.flags for p in pl):
        return None, pl
    pos, fl, ll, offs = segments.stitch_tables(pl)
    kept = [segments.keep_local(fr, p, o) for fr, p, o in zip(local, pl, offs)]
    return segments.merge_frames(kept, decode_table(make, pos, fl)), pl


def _ts33c_stream(rng, n_frames, rate=20000.0, held=False):
    sym = []
    for k in range(n_frames):
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
        if rng.random() < 0.4:
            sym += [(0, float(rng.integers(2000, 40000)))]
        if held and rng.random() < 0.4:
            sym += [(1, float(rng.integers(1600, 9000))), (0, float(rng.integers(1600, 5000)))]   # a held pulse: over-limit HIGH run
    env = synth.envelope(sym, rate, lead_in_us=float(rng.integers(0, 8000)))
    env = np.concatenate([env, np.zeros(int(rng.integers(0, 400)), np.uint8)])
    return env.astype(np.int8)


def _tfa_stream(rng, n_frames, rate=2e6):
    sym = []
    for _ in range(n_frames):
        pk = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10.0,
                              humidity=int(rng.integers(0, 100)))
        sym += synth.tfa_symbols(synth.tfa_bits(pk), rng, reset_gap_us=float(rng.integers(600, 3000)))
    env = synth.envelope(sym, rate, lead_in_us=float(rng.integers(100, 2000)))
    return np.concatenate([env, np.zeros(int(rng.integers(0, 3000)), np.uint8)]).astype(np.int8)


def _single(make, bits):
    o = make()
    o.work([bits])
    return [to_frame(f) for f in o.frames]


@pytest.mark.parametrize("seed", range(12))
def test_ts33c_frames_only_random_cuts(seed):
    rng = np.random.default_rng(1000 + seed)
    make = lambda: odec.TS33CStreamDecoder(sample_rate=20000)
    bits = _ts33c_stream(rng, int(rng.integers(3, 9)), held=seed % 2 == 1)
    want = _single(make, bits)
    assert len(want) >= 2
    for world in (2, 3, 5, 8):
        inner = np.sort(rng.integers(0, bits.size + 1, world - 1))
        cuts = [0] + [int(v) for v in inner] + [bits.size]
        got, pl = frames_by_segments(make, bits, cuts, 20000.0, 1550.0)
        assert got is not None
        assert [key(f) for f in got] == [key(f) for f in want], (world, cuts)


def test_cuts_at_every_position_of_a_short_stream():
    """every single cut position of a two-packet stream, and every pair on a coarse grid: a cut inside the
    preamble, inside a pulse, on an edge, inside the closing gap, one sample before / behind an over-limit run"""
    rng = np.random.default_rng(5)
    make = lambda: odec.TS33CStreamDecoder(sample_rate=20000)
    bits = _ts33c_stream(rng, 2)
    want = [key(f) for f in _single(make, bits)]
    assert len(want) == 2
    for c in range(0, bits.size + 1):
        got, _ = frames_by_segments(make, bits, [0, c, bits.size], 20000.0, 1550.0)
        assert [key(f) for f in got] == want, c
    grid = list(range(0, bits.size + 1, 97))
    for i, a in enumerate(grid):
        for b in grid[i::7]:
            got, _ = frames_by_segments(make, bits, [0, a, b, bits.size], 20000.0, 1550.0)
            assert [key(f) for f in got] == want, (a, b)


def test_all_zero_and_tiny_segments():
    make = lambda: odec.TS33CStreamDecoder(sample_rate=20000)
    z = np.zeros(5000, np.int8)
    got, _ = frames_by_segments(make, z, [0, 100, 2500, 5000], 20000.0, 1550.0)
    assert got == []
    rng = np.random.default_rng(8)
    bits = _ts33c_stream(rng, 3)
    want = [key(f) for f in _single(make, bits)]
    cuts = [0, 0, 1, 2, 40, 41, bits.size // 2, bits.size // 2, bits.size - 1, bits.size]
    got, _ = frames_by_segments(make, bits, cuts, 20000.0, 1550.0)
    assert [key(f) for f in got] == want


@pytest.mark.parametrize("seed", range(4))
def test_tfa_frames_only_random_cuts(seed):
    rng = np.random.default_rng(2000 + seed)
    make = lambda: _TFA(sample_rate=2000000)
    bits = _tfa_stream(rng, 4)
    want = _single(make, bits)
    assert len(want) >= 3
    for world in (2, 4, 8):
        inner = np.sort(rng.integers(0, bits.size + 1, world - 1))
        cuts = [0] + [int(v) for v in inner] + [bits.size]
        got, _ = frames_by_segments(make, bits, cuts, 2e6, 500.0)
        assert [key(f) for f in got] == [key(f) for f in want], (world, cuts)
    assert all(f.status == _lib.FRAME_OK for f in want)


def test_tables_that_do_not_fit_are_flagged():
    rng = np.random.default_rng(3)
    make = lambda: odec.TS33CStreamDecoder(sample_rate=20000)
    bits = _ts33c_stream(rng, 3)
    got, pl = frames_by_segments(make, bits, [0, bits.size // 2, bits.size], 20000.0, 1550.0, cap=16)
    assert got is None and any(p.flags for p in pl)
    with pytest.raises(ValueError):
        segments.stitch_tables(pl)


def test_pack_unpack_round_trip():
    p = segments.LocalPieces(1000, [0, 5, 9, 300], [700, 720, 1000], 1, 0, 300, 700, 0)
    q, k = segments.LocalPieces.unpack(p.pack(16, 7), 16)
    assert k == 7 and q.n_bits == 1000 and list(q.head) == [0, 5, 9, 300] and list(q.tail) == [700, 720, 1000]
    assert (q.first_level, q.last_level, q.keep_from, q.keep_to, q.flags) == (1, 0, 300, 700, 0)
    with pytest.raises(ValueError):
        p.pack(3)


# ---- the collective under gloo
def _capture_bits():
    rng = np.random.default_rng(77)
    return _ts33c_stream(rng, 7, held=True)


def _worker(rank, world, port, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bits = _capture_bits()
    a, b = segments.segment_bounds(bits.size, world, 1)[rank]
    seg = bits[a:b]
    make = lambda: odec.TS33CStreamDecoder(sample_rate=20000)
    carries, nbits = segments.gather_carries(seg.size, 0, 1, int(seg[-1]) if seg.size else 0, world)
    assert nbits[rank] == seg.size and len(carries) == world
    p = pieces_model(seg, 20000.0, 1550.0, cap)
    o = make()
    o.work([seg])
    frames, stats = segments.gather_frames(p, [to_frame(f) for f in o.frames], world,
                                           lambda pos, fl, ll: decode_table(make, pos, fl), cap=cap)
    q.put((rank, None if frames is None else [key(f) for f in frames], stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cap", [(2, 256), (3, 256), (3, 8)])
def test_gather_frames_gloo(world, cap):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, frames, stats = q.get(timeout=180)
        got[r] = (frames, stats)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bits = _capture_bits()
    want = [key(f) for f in _single(lambda: odec.TS33CStreamDecoder(sample_rate=20000), bits)]
    assert len(want) >= 5
    for r in range(world):
        frames, stats = got[r]
        if cap < 64:
            assert frames is None and stats["fallback"]      # every rank learns it from the headers
            continue
        assert frames == want, r
        assert stats["payload_bytes"] < 64 * 1024            # what a rank contributes to the two all_gathers
        assert stats["local_frames"] + stats["stitched_frames"] == len(want)
