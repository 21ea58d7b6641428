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
