"""One long capture cut into contiguous segments, one per GPU (SURVEY.md 8e;
BASELINE.json north_star: "each GPU gets a contiguous segment with FIR-history
overlap").

What crosses a segment boundary in the reference's single stream
(TS33C/live_decoder.grc:3292-3377 run from stream start) and how it is restored:

* FIR history / discriminator history: the ``history_len()`` samples in front of
  the segment (the halo) are handed to ``Chain.seek``;
* NCO phase, decimation phase, keep_one_in_n counter: pure functions of the
  absolute sample index, set by ``Chain.seek``;
* threshold_ff state: every segment is computed with carry-in ``thr_init`` and
  reports (has_event, last_state, n_undetermined); ``resolve_carries`` is the
  exclusive scan over segments, ``patch`` rewrites the undetermined prefix;
* decoder state (rawBuffer, currentSymbols, ...): the stitched sliced stream
  (1/(D*keep_n) byte per input sample) is gathered and decoded by ONE decoder
  object, exactly as ``StreamDecoder.work([stream])`` sees it
  (TS33C/stream_decoder.py:39-220).

The collective is an all_gather of four int64 per rank plus one of the padded
sliced bytes; "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

Frames-only form (north_star: "RCCL ... used only to gather decoded frames and
counts"; `frames_only`, `gather_frames` below).  The stream decoders cut the
symbol widths behind every width above reset_limit and decode each closed piece
from an empty currentSymbols (TS33C/stream_decoder.py:76-110), so a piece that
lies wholly inside a segment decodes there to what the single stream gives.
Each rank therefore decodes its own segment (sdr_segment_decode) and hands out
only: its frames, and the run starts of its "head" (up to the end of its first
over-limit run) and "tail" (behind its last over-limit run that does not touch
the segment's end).  Joined over the ranks -- a seam is a run start only where
the levels on its two sides differ -- these form a short run table in which
everything a rank decoded by itself has collapsed into one over-limit run; ONE
work() call on that table (sdr_process_runs) yields the frames of the pieces
that touch a boundary, cross-chunk rules (:88-105,212) included because it IS
the single stream's call with the closed interior pieces cut out.
"""
import numpy as np


def segment_bounds(n_total, world, align=1):
    """[(start, end)] of `world` contiguous segments; interior boundaries are
    multiples of `align` (use decim*keep_n so every segment emits whole bytes)."""
    per = -(-n_total // world)
    per = -(-per // align) * align
    out = []
    for r in range(world):
        a = min(n_total, r * per)
        out.append((a, min(n_total, a + per)))
    return out


def resolve_carries(has_event, last_state, thr_init=0):
    """Exclusive scan of the hysteresis state over segments -> carry-in of each."""
    carries, c = [], int(thr_init)
    for he, ls in zip(has_event, last_state):
        carries.append(c)
        if he:
            c = int(ls)
    return carries


def patch(bits, n_undetermined, carry_in, computed_with=0):
    """In place: the segment's leading `n_undetermined` bytes hold the state the
    stream had when the segment began.  Works on numpy arrays and torch tensors."""
    if carry_in != computed_with and n_undetermined:
        bits[:int(n_undetermined)] = carry_in
    return bits


def stitch_segments(bits_list, infos, thr_init=0):
    """Single-process form (tests, one GPU running the segments in turn):
    -> the capture's whole sliced stream as one int8 array."""
    carries = resolve_carries([i.has_event for i in infos], [i.last_state for i in infos], thr_init)
    parts = []
    for b, i, c in zip(bits_list, infos, carries):
        b = np.array(b, np.int8, copy=True)
        parts.append(patch(b, i.n_undetermined, c, thr_init))
    return np.concatenate(parts) if parts else np.zeros(0, np.int8)


def gather_stream(bits, n_bits, n_undetermined, has_event, last_state, world, thr_init=0):
    """Collective form, one rank per segment (rank order == segment order).
    bits: uint8/int8 torch tensor (this rank's device) whose first n_bits bytes
    are the segment's sliced stream computed with carry-in `thr_init`.
    Returns (stream, offsets): the whole capture's sliced stream on every rank
    and each segment's start offset in it."""
    import torch
    import torch.distributed as dist
    dev = bits.device
    mine = torch.tensor([int(n_bits), int(n_undetermined), int(has_event), int(last_state)], dtype=torch.int64,
                        device=dev)
    allv = torch.zeros(4 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allv, mine)
    tab = allv.view(world, 4).cpu().numpy()
    carries = resolve_carries(tab[:, 2], tab[:, 3], thr_init)
    rank = dist.get_rank()
    patch(bits, n_undetermined, carries[rank], thr_init)
    width = int(tab[:, 0].max()) if world else 0
    width = max(16, -(-width // 16) * 16)
    send = torch.zeros(width, dtype=bits.dtype, device=dev)
    send[:int(n_bits)] = bits[:int(n_bits)]
    recv = torch.zeros(world * width, dtype=bits.dtype, device=dev)
    dist.all_gather_into_tensor(recv, send)
    parts, offsets, off = [], [], 0
    for r in range(world):
        k = int(tab[r, 0])
        parts.append(recv[r * width:r * width + k])
        offsets.append(off)
        off += k
    return torch.cat(parts), offsets


def stitch_rows(bits_list, infos_list, thr_init=0):
    """Single-process form for a multi-row handle (channelizer: 64 channel rows per segment).
    bits_list[s]: (rows, n_s) array of segment s; infos_list[s]: its per-row SegmentInfo list.
    -> (rows, total) int8: every row's stream over the whole capture."""
    rows = len(infos_list[0])
    out = []
    for r in range(rows):
        carries = resolve_carries([inf[r].has_event for inf in infos_list], [inf[r].last_state for inf in infos_list], thr_init)
        parts = []
        for b, inf, c in zip(bits_list, infos_list, carries):
            row = np.array(np.asarray(b)[r], np.int8, copy=True)
            parts.append(patch(row, inf[r].n_undetermined, c, thr_init))
        out.append(np.concatenate(parts))
    return np.stack(out)


def gather_rows(bits, n_bits, n_undetermined, has_event, last_state, world, thr_init=0):
    """Collective form for a multi-row handle, one rank per segment (rank order == segment order).
    bits: (rows, >= n_bits) uint8/int8 torch tensor on this rank's device, computed with carry-in `thr_init`;
    n_undetermined / has_event / last_state: one value per row (sdr_segment_info_rows).
    Two all_gathers: 3 int64 per row + the segment length, then the padded rows.
    Returns (stream[rows, total], offsets) on every rank."""
    import torch
    import torch.distributed as dist
    dev = bits.device
    rows = int(bits.shape[0])
    mine = torch.zeros(3 * rows + 1, dtype=torch.int64, device=dev)
    mine[0] = int(n_bits)
    mine[1:1 + rows] = torch.as_tensor([int(v) for v in n_undetermined], dtype=torch.int64)
    mine[1 + rows:1 + 2 * rows] = torch.as_tensor([int(v) for v in has_event], dtype=torch.int64)
    mine[1 + 2 * rows:] = torch.as_tensor([int(v) for v in last_state], dtype=torch.int64)
    allv = torch.zeros(world * mine.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allv, mine)
    tab = allv.view(world, -1).cpu().numpy()
    nb = tab[:, 0]
    und, he, ls = tab[:, 1:1 + rows], tab[:, 1 + rows:1 + 2 * rows], tab[:, 1 + 2 * rows:]
    rank = dist.get_rank()
    for r in range(rows):
        carries = resolve_carries(he[:, r], ls[:, r], thr_init)
        patch(bits[r], und[rank, r], carries[rank], thr_init)
    width = int(nb.max()) if world else 0
    width = max(16, -(-width // 16) * 16)
    send = torch.zeros((rows, width), dtype=bits.dtype, device=dev)
    send[:, :int(n_bits)] = bits[:, :int(n_bits)]
    recv = torch.zeros(world * rows * width, dtype=bits.dtype, device=dev)
    dist.all_gather_into_tensor(recv, send.reshape(-1))
    recv = recv.view(world, rows, width)
    parts, offsets, off = [], [], 0
    for w in range(world):
        k = int(nb[w])
        parts.append(recv[w, :, :k])
        offsets.append(off)
        off += k
    return torch.cat(parts, dim=1), offsets


def run_segment(chain, iq, start, end):
    """Front end of ONE segment of a capture that this process can index as
    `iq[a:b]` (numpy array or torch tensor on the GPU): seek with the halo, one
    process_iq call, -> SegmentInfo (bits stay in the handle: fetch_bits /
    export_bits)."""
    H = chain.history_len()
    a = max(0, start - H)
    chain.seek(start, iq[a:start] if start > a else None)
    chain.process_iq(iq[start:end])
    return chain.segment_info()


# ------------------------------------------------------------------ frames-only exchange
class LocalPieces(object):
    """What sdr_segment_decode leaves of a segment (include/sdrhip.h: sdr_segment_pieces): run starts of the head
    and the tail counted from the segment's first byte, the keep window of the segment's own frames, the levels of
    its first and last byte, flags (!= 0: use the byte exchange for this capture)."""
    __slots__ = ("n_bits", "head", "tail", "first_level", "last_level", "keep_from", "keep_to", "flags")
    HDR = 8

    def __init__(self, n_bits, head, tail, first_level, last_level, keep_from, keep_to, flags=0):
        self.n_bits = int(n_bits)
        self.head = np.asarray(head, np.int64)
        self.tail = np.asarray(tail, np.int64)
        self.first_level = int(first_level)
        self.last_level = int(last_level)
        self.keep_from = int(keep_from)
        self.keep_to = int(keep_to)
        self.flags = int(flags)

    def pack(self, cap, n_frames=0):
        """fixed-size int64 record for the all_gather: 8 header words + head[cap] + tail[cap]"""
        if self.head.size > cap or self.tail.size > cap:
            raise ValueError("head / tail longer than the record's capacity")
        v = np.zeros(self.HDR + 2 * cap, np.int64)
        v[:8] = [self.n_bits, self.head.size, self.tail.size, self.first_level, self.last_level, self.keep_from,
                 self.keep_to, self.flags | (int(n_frames) << 8)]
        v[8:8 + self.head.size] = self.head
        v[8 + cap:8 + cap + self.tail.size] = self.tail
        return v

    @classmethod
    def unpack(cls, v, cap):
        """-> (LocalPieces, n_frames)"""
        v = np.asarray(v, np.int64)
        nh, nt = int(v[1]), int(v[2])
        return (cls(v[0], v[8:8 + nh], v[8 + cap:8 + cap + nt], v[3], v[4], v[5], v[6], int(v[7]) & 0xff),
                int(v[7]) >> 8)


def stitch_tables(pieces_list):
    """Join the segments' heads and tails into the run table of the whole capture with every locally decoded
    stretch collapsed into one run.  -> (pos int32[], first_level, last_level, offsets): pos[0] = 0 < ... <
    pos[-1] = the capture's sliced length (what sdr_process_runs takes), offsets[s] = first byte of segment s."""
    pos, offsets, off = [0], [], 0
    first_level = last_level = None
    for p in pieces_list:
        offsets.append(off)
        if p.n_bits == 0:
            continue
        if p.flags:
            raise ValueError("segment flagged %d: its tables are incomplete, use the byte exchange" % p.flags)
        local = np.concatenate([p.head, p.tail]) if p.tail.size else p.head
        if local.size < 2 or local[0] != 0 or local[-1] != p.n_bits:
            raise ValueError("segment table must run from 0 to n_bits")
        if first_level is None:
            first_level = p.first_level
        elif p.first_level != last_level:
            pos.append(off)                       # the seam is an edge of the stream
        inner = local[1:-1] + off
        pos.extend(int(x) for x in inner)
        last_level = p.last_level
        off += p.n_bits
    pos.append(off)
    out = np.unique(np.asarray(pos, np.int64))    # head's last entry == tail's first when nothing lies between
    if off > 0x7fffffff:
        raise ValueError("capture longer than 2^31 sliced bytes")
    if off == 0:
        return np.zeros(0, np.int32), 0, 0, offsets
    return out.astype(np.int32), int(first_level), int(last_level), offsets


def keep_local(frames, pieces, offset):
    """The frames of a segment's own pass that are final (closed inside the segment, behind its head and in front
    of its tail), with end_pos moved to the capture's coordinates.  `frames`: objects with an end_pos attribute
    (_lib.Frame); they are modified in place."""
    out = []
    for f in frames:
        if pieces.keep_from < f.end_pos <= pieces.keep_to:
            f.end_pos = int(f.end_pos) + int(offset)
            out.append(f)
    return out


def merge_frames(local_lists, stitched):
    """-> all frames in stream order (a frame's end_pos is the end of the over-limit width that closed it: unique)."""
    allf = [f for lst in local_lists for f in lst] + list(stitched)
    allf.sort(key=lambda f: int(f.end_pos))
    return allf


def frames_only(chain, iq, bounds, cap=1024):
    """Single-process form (tests, one GPU running the segments in turn): the capture's frames without ever joining
    the segments' bytes.  -> (frames, stats); stats["fallback"] is True when a table did not fit `cap` and the byte
    exchange was used instead."""
    carry = int(chain.cfg.thr_init)
    pieces, locals_ = [], []
    for a, b in bounds:
        info = run_segment(chain, iq, a, b)
        p = chain.segment_decode(carry, cap)
        fr = chain.fetch_frames() if p.n_bits else []
        pieces.append(p)
        locals_.append(fr)
        if info.has_event:
            carry = int(info.last_state)
    stats = {"fallback": any(p.flags for p in pieces), "table_entries": sum(p.head.size + p.tail.size for p in pieces)}
    if stats["fallback"]:
        return None, stats
    pos, fl, ll, offs = stitch_tables(pieces)
    kept = [keep_local(fr, p, o) for fr, p, o in zip(locals_, pieces, offs)]
    stitched = []
    if pos.size >= 2:
        chain.seek(0)    # a fresh decoder object: state idle, no pending symbols
        try:
            chain.process_runs(pos, fl, ll)
            stitched = chain.fetch_frames()
        except Exception as e:   # shorter than reset_limit: the reference would only buffer it
            if "shorter than reset_limit" not in str(e):
                raise
    stats["local_frames"] = sum(len(k) for k in kept)
    stats["stitched_frames"] = len(stitched)
    stats["runs"] = int(pos.size) - 1
    return merge_frames(kept, stitched), stats


def gather_frames(pieces, frames, world, decode_runs, cap=1024, device="cpu"):
    """Collective form of `frames_only`, one rank per segment (rank order == segment order); call it behind
    run_segment + the 4-int64 carry scan (`gather_carries`) + Chain.segment_decode.  pieces / frames: this rank's
    LocalPieces and ALL frames of its local pass (the keep window is applied here).  decode_runs(pos, first_level,
    last_level) -> frames of ONE work() call of a fresh decoder on the joined table (Chain.seek(0) +
    Chain.process_runs + fetch_frames); every rank runs it on the same few hundred runs, so every rank ends with
    the capture's frames and no broadcast follows.
    Two all_gathers: one fixed int64 record per rank (8 + 2*cap words), then the kept frame records padded to the
    largest count.  -> (frames, stats); frames is None when some rank's tables did not fit (stats["fallback"]):
    the caller then falls back to `gather_stream`."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import _lib
    FB = C.sizeof(_lib.Frame)
    mine = [f for f in frames if pieces.keep_from < f.end_pos <= pieces.keep_to] if not pieces.flags else []
    if pieces.flags:   # incomplete tables are not sent: the header alone tells every rank to fall back
        pieces = LocalPieces(pieces.n_bits, [], [], pieces.first_level, pieces.last_level, 0, 0, pieces.flags)
    rec = torch.from_numpy(pieces.pack(cap, len(mine))).to(device)
    allrec = torch.zeros(world * rec.numel(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allrec, rec)
    tab = allrec.view(world, -1).cpu().numpy()
    plist, counts = [], []
    for r in range(world):
        p, k = LocalPieces.unpack(tab[r], cap)
        plist.append(p)
        counts.append(k)
    stats = {"fallback": any(p.flags for p in plist), "payload_bytes": int(rec.numel()) * 8,
             "table_entries": int(sum(p.head.size + p.tail.size for p in plist))}
    if stats["fallback"]:
        return None, stats
    pos, fl, ll, offs = stitch_tables(plist)
    rank = dist.get_rank()
    kmax = max(1, max(counts))
    send = np.zeros(kmax * FB, np.uint8)
    for i, f in enumerate(mine):
        f.end_pos = int(f.end_pos) + offs[rank]
        send[i * FB:(i + 1) * FB] = np.frombuffer(bytes(f), np.uint8)
    send_t = torch.from_numpy(send).to(device)
    recv = torch.zeros(world * kmax * FB, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send_t)
    stats["payload_bytes"] += int(send_t.numel())
    raw = recv.cpu().numpy()
    kept = []
    for r in range(world):
        kept.append([_lib.Frame.from_buffer_copy(raw[(r * kmax + i) * FB:(r * kmax + i + 1) * FB].tobytes())
                     for i in range(counts[r])])
    stitched = decode_runs(pos, fl, ll) if pos.size >= 2 else []
    stats["local_frames"] = int(sum(counts))
    stats["stitched_frames"] = len(stitched)
    stats["runs"] = max(0, int(pos.size) - 1)
    return merge_frames(kept, stitched), stats


def gather_carries(n_bits, n_undetermined, has_event, last_state, world, thr_init=0, device="cpu"):
    """The 4-int64 all_gather of `gather_stream` by itself: -> (carry-in of every segment, n_bits of every segment)."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([int(n_bits), int(n_undetermined), int(has_event), int(last_state)], dtype=torch.int64, device=device)
    allv = torch.zeros(4 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allv, mine)
    tab = allv.view(world, 4).cpu().numpy()
    return resolve_carries(tab[:, 2], tab[:, 3], thr_init), [int(v) for v in tab[:, 0]]
