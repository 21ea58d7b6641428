"""Receive chain = what one GRC flowgraph wires between the IQ source and the
decoder block, as one object backed by one libsdrhip handle.

Parameter names follow the GRC blocks: ``low_pass(...)`` is the
``low_pass_filter`` block's taps (firdes.low_pass), ``Chain(decim=, center_freq=,
taps=)`` the freq_xlating_fir_filter / fir_filter, ``thr=(low, high)`` the
threshold block, ``keep_n`` keep_one_in_n (ref: TS33C/live_decoder.grc:3292-3377).
Inputs may be numpy arrays (copied to the GPU by the library) or torch tensors
already in HBM (passed by device pointer, no copy)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (DEC_BALDR, DEC_BERNER, DEC_KWMOBILE, DEC_MUMBI, DEC_VELUX, DEC_NONE, DEC_SOMFY, DEC_TFA, DEC_TFA_GRC, DEC_TS33C,  # noqa: F401
                   DEMOD_MAG2, DEMOD_NONE, DEMOD_QUAD, WIN_BLACKMAN, WIN_HAMMING, WIN_KAISER)


def low_pass(gain, fs, fc, tw, win=WIN_HAMMING, beta=6.76):
    """firdes.low_pass(gain, sampling_freq, cutoff_freq, transition_width, window, beta)."""
    L = _lib.load()
    buf = (C.c_float * 65536)()
    n = L.sdr_firdes_low_pass(gain, fs, fc, tw, win, beta, buf, 65536)
    if n <= 0:
        raise _lib.SdrError(n, "sdr_firdes_low_pass")
    return np.frombuffer(buf, np.float32, n).copy()


def resampler_taps(interp, decim, fractional_bw=0.4):
    """rational_resampler's design_filter (empty `taps` in GRC)."""
    L = _lib.load()
    buf = (C.c_float * 65536)()
    n = L.sdr_resampler_taps(interp, decim, fractional_bw, buf, 65536)
    if n <= 0:
        raise _lib.SdrError(n, "sdr_resampler_taps")
    return np.frombuffer(buf, np.float32, n).copy()


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Chain(object):
    def __init__(self, sample_rate, decim=1, center_freq=0.0, taps=None, demod=DEMOD_MAG2, demod_gain=1.0,
                 thr=(0.05, 0.15), thr_init=0, keep_n=1, decoder=DEC_NONE, decoder_rate=0.0, post_taps=None,
                 tap_demod=False, max_batch=1, max_frames=16, max_samples=1 << 20, squelch=None,
                 pre_taps=None, pre_decim=1, input_format="fc32", channels=0):
        """squelch=(threshold_db, alpha[, gate=True]): analog_pwr_squelch_cc between the channel
        filter and the demodulator (TFA/decoder.grc:391-446).
        pre_taps/pre_decim: an upstream fir_filter_ccf in front of this chain's own filter
        (mumbi/FS300.grc: low_pass_filter -> rational_resampler 1/10); sample_rate and
        max_samples are then those of the upstream input.
        input_format: "fc32" (complex64) or "cu8" (rtl-sdr bytes, uint8 I/Q pairs converted as
        rtlsdr_source does; arrays of shape (..., n, 2) or (..., 2n) uint8).
        channels=64: polyphase channelizer handle (sdr_pfb_create): `taps` is the prototype
        low-pass, decimation is 64, batch rows come back as capture*64 + channel."""
        self._L = _lib.load()
        self._cu8 = {"fc32": False, "cu8": True}[input_format]
        self._h = C.c_void_p()
        cfg = _lib.ChainCfg()
        pre_decim = int(pre_decim) if pre_taps is not None else 1
        cfg.sample_rate = float(sample_rate) / pre_decim
        max_samples = -(-int(max_samples) // pre_decim)
        cfg.center_freq = float(center_freq)
        cfg.decim = int(decim)
        cfg.demod = int(demod)
        cfg.demod_gain = float(demod_gain)
        cfg.thr_lo, cfg.thr_hi = float(thr[0]), float(thr[1])
        cfg.thr_init = int(thr_init)
        cfg.keep_n = int(keep_n)
        cfg.decoder = int(decoder)
        cfg.decoder_rate = float(decoder_rate)
        cfg.post_ntaps = 0 if post_taps is None else int(len(post_taps))
        cfg.tap_demod = int(bool(tap_demod))
        cfg.max_batch = int(max_batch)
        cfg.max_frames = int(max_frames)
        cfg.max_samples = int(max_samples)
        self.cfg = cfg
        self.channels = int(channels)
        if self.channels:
            rc = self._L.sdr_pfb_create(C.byref(cfg), self.channels, C.byref(self._h))
        else:
            rc = self._L.sdr_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            _lib.check(None, rc)
        if taps is not None:
            self.set_taps(taps)
        if post_taps is not None:
            pt = np.ascontiguousarray(post_taps, np.float32)
            _lib.check(self._h, self._L.sdr_set_post_taps(self._h, pt.ctypes.data_as(C.POINTER(C.c_float)), pt.size))
        self.out = _lib.Out()
        if pre_taps is not None:
            pt = np.ascontiguousarray(pre_taps, np.float32)
            _lib.check(self._h, self._L.sdr_set_prefilter(self._h, pt.ctypes.data_as(C.POINTER(C.c_float)), pt.size,
                                                          pre_decim))
        if squelch is not None:
            self.set_squelch(*squelch)
        if self._cu8:
            _lib.check(self._h, self._L.sdr_set_input_format(self._h, _lib.IN_CU8))

    def set_squelch(self, threshold_db, alpha, gate=True):
        _lib.check(self._h, self._L.sdr_set_squelch(self._h, float(threshold_db), float(alpha), int(bool(gate))))

    def fetch_squelch_counts(self):
        """Items that passed the squelch in the last call, per capture."""
        n = C.c_size_t()
        _lib.check(self._h, self._L.sdr_fetch_squelch_counts(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        if n.value:
            _lib.check(self._h, self._L.sdr_fetch_squelch_counts(self._h, C.c_void_p(out.ctypes.data), out.size,
                                                                 C.byref(n)))
        return out

    # -- life cycle
    def close(self):
        if self._h:
            self._L.sdr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_taps(self, taps):
        taps = np.asarray(taps)
        if np.iscomplexobj(taps):
            t = np.ascontiguousarray(taps, np.complex64)
            _lib.check(self._h, self._L.sdr_set_taps(self._h, t.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)), t.size, 1))
        else:
            t = np.ascontiguousarray(np.atleast_1d(taps), np.float32)
            _lib.check(self._h, self._L.sdr_set_taps(self._h, t.ctypes.data_as(C.POINTER(C.c_float)), t.size, 0))

    def set_stream(self, hip_stream_ptr):
        _lib.check(self._h, self._L.sdr_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def synchronize(self):
        _lib.check(self._h, self._L.sdr_synchronize(self._h))

    def set_overlap(self, on=True):
        """sdr_set_overlap: run the back end of each batch (and redecode*/export_*) on a second stream beside
        the next batch's front end.  Results are the same; device-side exports are ordered on result_stream()."""
        _lib.check(self._h, self._L.sdr_set_overlap(self._h, int(bool(on))))

    def result_stream(self):
        """hipStream_t (int) that device-side results are ordered on (torch.cuda.ExternalStream takes it)."""
        return int(self._L.sdr_result_stream(self._h) or 0)

    def overlap_state(self):
        """sdr_overlap_state: non-blocking look at the handle's two streams and their events (8 ints, see
        include/sdrhip.h); callable from another thread while a call on the handle blocks."""
        st = (C.c_int32 * 8)()
        _lib.check(self._h, self._L.sdr_overlap_state(self._h, C.byref(st)))
        return list(st)

    # -- data path
    def _nsamp(self, keep, dev):
        """complex samples in a 1-D stream argument"""
        n = int(keep.numel() if dev else keep.size)
        if self._cu8:
            return n // 2
        if dev and not keep.is_complex():
            n //= 2
        return n

    def _iq_ptr(self, iq):
        if self._cu8:
            if _is_torch(iq) and iq.is_cuda:
                if not iq.is_contiguous() or iq.element_size() != 1:
                    raise ValueError("cu8 tensor must be contiguous uint8")
                return iq.data_ptr(), 1, iq
            a = np.ascontiguousarray(iq.cpu().numpy() if _is_torch(iq) else iq, np.uint8)
            return a.ctypes.data, 0, a
        if _is_torch(iq):
            if not iq.is_cuda:
                iq = iq.cpu().numpy()
            else:
                if not iq.is_contiguous():
                    raise ValueError("IQ tensor must be contiguous")
                return iq.data_ptr(), 1, iq
        a = np.ascontiguousarray(iq, np.complex64)
        return a.ctypes.data, 0, a

    def process_iq(self, iq):
        """One streaming call (state carried): n complex64 samples."""
        ptr, dev, keep = self._iq_ptr(iq)
        n = self._nsamp(keep, dev)
        _lib.check(self._h, self._L.sdr_process_iq(self._h, C.c_void_p(ptr), n, dev, C.byref(self.out)))
        return self.out

    def process_bits(self, bits):
        if _is_torch(bits) and bits.is_cuda:
            if not bits.is_contiguous() or bits.element_size() != 1:
                raise ValueError("bit tensor must be contiguous int8/uint8")
            _lib.check(self._h, self._L.sdr_process_bits(self._h, C.c_void_p(bits.data_ptr()), bits.numel(), 1,
                                                         C.byref(self.out)))
            return self.out
        b = np.ascontiguousarray(bits, np.int8)
        _lib.check(self._h, self._L.sdr_process_bits(self._h, C.c_void_p(b.ctypes.data), b.size, 0, C.byref(self.out)))
        return self.out

    def batch_process(self, iq):
        """iq: (n_captures, n_per_capture) complex64 (numpy, or torch on the GPU;
        a float32 torch tensor of shape (n_captures, n, 2) is accepted too)."""
        ptr, dev, keep = self._iq_ptr(iq)
        ncap = int(keep.shape[0])
        n = int(keep.shape[1])
        if self._cu8 and len(keep.shape) == 2:
            n //= 2   # (ncap, 2n) bytes
        _lib.check(self._h, self._L.sdr_batch_process(self._h, C.c_void_p(ptr), n, n, ncap, dev, C.byref(self.out)))
        return self.out

    def batch_process_static(self, iq, bits_out, frames_out, counts_out):
        """sdr_batch_process_static: the allocation-free variant.  All four arguments are torch tensors on
        this GPU: iq as for batch_process, bits_out uint8/int8 [>= ncap*n_bits], frames_out uint8
        [>= ncap*max_frames*96], counts_out int32 [>= ncap]."""
        if not (_is_torch(iq) and iq.is_cuda and iq.is_contiguous()):
            raise ValueError("batch_process_static takes a contiguous tensor in device memory")
        ncap, n = int(iq.shape[0]), int(iq.shape[1])
        if self._cu8 and iq.dim() == 2:
            n //= 2
        if counts_out.numel() < ncap:
            raise ValueError("counts_out too small")
        _lib.check(self._h, self._L.sdr_batch_process_static(
            self._h, C.c_void_p(iq.data_ptr()), n, n, ncap, C.c_void_p(bits_out.data_ptr()),
            bits_out.numel() * bits_out.element_size(), C.c_void_p(frames_out.data_ptr()),
            frames_out.numel() * frames_out.element_size() // C.sizeof(_lib.Frame), C.c_void_p(counts_out.data_ptr()),
            C.byref(self.out)))
        return self.out

    # -- results
    def fetch_frames(self):
        n = C.c_size_t()
        rc = self._L.sdr_fetch_frames(self._h, None, 0, C.byref(n))
        _lib.check(self._h, rc)
        if n.value == 0:
            return []
        buf = (_lib.Frame * n.value)()
        _lib.check(self._h, self._L.sdr_fetch_frames(self._h, buf, n.value, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def fetch_bits(self):
        n = C.c_size_t()
        _lib.check(self._h, self._L.sdr_fetch_bits(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int8)
        if n.value:
            _lib.check(self._h, self._L.sdr_fetch_bits(self._h, C.c_void_p(out.ctypes.data), out.size, C.byref(n)))
        return out

    def fetch_demod(self):
        n = C.c_size_t()
        _lib.check(self._h, self._L.sdr_fetch_demod(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            _lib.check(self._h, self._L.sdr_fetch_demod(self._h, C.c_void_p(out.ctypes.data), out.size, C.byref(n)))
        return out.view(np.complex64) if self.cfg.demod == DEMOD_NONE else out

    # -- one long capture in segments (sdr_seek / sdr_segment_info_get)
    def history_len(self):
        """Input samples in front of a segment that seek() needs: the channel filter's history, or -- for a
        chain with a prefilter stage or a post-demod filter -- the sum of the stages' spans."""
        return int(self._L.sdr_history_len(self._h))

    def seek(self, abs_index, halo=None):
        """Position the streaming state at sample `abs_index` of a capture; `halo`
        = the samples that precede it (at least min(abs_index, history_len()))."""
        if halo is None:
            _lib.check(self._h, self._L.sdr_seek(self._h, int(abs_index), None, 0, 0))
            return
        ptr, dev, keep = self._iq_ptr(halo)
        n = self._nsamp(keep, dev)
        _lib.check(self._h, self._L.sdr_seek(self._h, int(abs_index), C.c_void_p(ptr), n, dev))

    def segment_info(self):
        info = _lib.SegmentInfo()
        _lib.check(self._h, self._L.sdr_segment_info_get(self._h, C.byref(info)))
        return info

    def segment_info_rows(self):
        """per-row summaries of the last call (channelizer: one per channel)"""
        n = C.c_size_t()
        _lib.check(self._h, self._L.sdr_segment_info_rows(self._h, None, 0, C.byref(n)))
        buf = (_lib.SegmentInfo * max(1, n.value))()
        if n.value:
            _lib.check(self._h, self._L.sdr_segment_info_rows(self._h, buf, n.value, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def batch_process_bits(self, bits):
        """Decoder only: rows of sliced bytes, shape (rows, n) -- numpy int8/uint8 or a torch tensor on the GPU --
        each decoded by a fresh decoder as one work() call; then fetch_frames()."""
        if _is_torch(bits) and bits.is_cuda:
            if bits.dim() != 2 or bits.element_size() != 1 or bits.stride(1) != 1:
                raise ValueError("bits tensor must be 2-D int8/uint8 with contiguous rows")
            _lib.check(self._h, self._L.sdr_batch_process_bits(self._h, C.c_void_p(bits.data_ptr()), int(bits.shape[1]),
                                                               int(bits.stride(0)), int(bits.shape[0]), 1, C.byref(self.out)))
            return self.out
        b = np.ascontiguousarray(bits, np.int8)
        if b.ndim != 2:
            raise ValueError("bits must be 2-D (rows, n)")
        _lib.check(self._h, self._L.sdr_batch_process_bits(self._h, C.c_void_p(b.ctypes.data), b.shape[1], b.shape[1],
                                                           b.shape[0], 0, C.byref(self.out)))
        return self.out

    def segment_decode(self, carry_in, cap=512):
        """sdr_segment_decode: decode the segment of the last process_iq call by itself, given the threshold carry-in
        the scan over the segments' summaries assigned to it.  -> segments.LocalPieces (head / tail run starts
        counted from the segment's first byte, the keep window for the local frames, levels, flags)."""
        from . import segments
        head = np.zeros(cap, np.int32)
        tail = np.zeros(cap, np.int32)
        info = _lib.SegmentPieces()
        _lib.check(self._h, self._L.sdr_segment_decode(self._h, int(carry_in), C.c_void_p(head.ctypes.data), cap,
                                                       C.c_void_p(tail.ctypes.data), cap, C.byref(info)))
        return segments.LocalPieces(int(info.n_bits), head[:info.n_head].copy(), tail[:info.n_tail].copy(),
                                    int(info.first_level), int(info.last_level), int(info.keep_from), int(info.keep_to),
                                    int(info.flags))

    def process_runs(self, pos, first_level, last_level):
        """sdr_process_runs: ONE work() call of the stream decoder on a stream given by its run starts
        (pos[0] = 0 ... pos[-1] = length) and its first / last sample's level; then fetch_frames()."""
        p = np.ascontiguousarray(pos, np.int32)
        _lib.check(self._h, self._L.sdr_process_runs(self._h, C.c_void_p(p.ctypes.data), p.size, int(first_level),
                                                     int(last_level), C.byref(self.out)))
        return self.out

    def export_bits(self, bits_tensor):
        """Last call's sliced bytes into a torch uint8/int8 tensor on this GPU; -> count."""
        n = C.c_size_t()
        _lib.check(self._h, self._L.sdr_export_bits(self._h, C.c_void_p(bits_tensor.data_ptr()),
                                                    bits_tensor.numel() * bits_tensor.element_size(), C.byref(n)))
        return n.value

    def redecode(self, decoder, decoder_rate=0.0):
        """Run another decoder over the last batch's sliced bytes; then fetch_frames()."""
        _lib.check(self._h, self._L.sdr_redecode(self._h, int(decoder), float(decoder_rate)))

    def redecode_multi(self, decoders, decoder_rate=0.0):
        """Run several decoders over the last batch's sliced bytes in ONE launch -> [frames of decoders[0], ...]."""
        arr = (C.c_int32 * len(decoders))(*[int(d) for d in decoders])
        _lib.check(self._h, self._L.sdr_redecode_multi(self._h, arr, len(decoders), float(decoder_rate)))
        return self

    def fetch_frames_multi(self, k):
        n = C.c_size_t()
        _lib.check(self._h, self._L.sdr_fetch_frames_multi(self._h, int(k), None, 0, C.byref(n)))
        if n.value == 0:
            return []
        buf = (_lib.Frame * n.value)()
        _lib.check(self._h, self._L.sdr_fetch_frames_multi(self._h, int(k), buf, n.value, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def decoder_state(self):
        st = (C.c_int32 * 4)()
        _lib.check(self._h, self._L.sdr_decoder_state(self._h, C.byref(st)))
        return list(st)

    def _fetch_array(self, fn, dtype, *lead):
        n = C.c_size_t()
        _lib.check(self._h, fn(self._h, *lead, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype)
        if n.value:
            _lib.check(self._h, fn(self._h, *lead, C.c_void_p(out.ctypes.data), out.size, C.byref(n)))
        return out

    def fetch_pending_symbols(self):
        """`currentSymbols` of the decoder object: float64 widths in microseconds (TS33C/stream_decoder.py:27,102)"""
        return self._fetch_array(self._L.sdr_fetch_pending_symbols, np.float64)

    def fetch_raw_buffer(self):
        """`rawBuffer` of the decoder object: the bytes held back so far (TS33C/stream_decoder.py:28,44-47)"""
        return self._fetch_array(self._L.sdr_fetch_raw_buffer, np.int8)

    def fetch_runs(self, row=0):
        """run table of the last decoding call: runs + 1 positions (empty if it decoded nothing)"""
        return self._fetch_array(self._L.sdr_fetch_runs, np.int32, int(row))

    def profile_enable(self, on=True, every=1):
        """HIP-event timing of the front-end / back-end launches: of every call, or of every `every`-th one."""
        _lib.check(self._h, self._L.sdr_profile_enable(self._h, int(every) if on else 0))

    def profile_read(self):
        """-> (front_ms[], back_ms[]) HIP-event durations of the last <=256 calls."""
        f = (C.c_float * 256)()
        b = (C.c_float * 256)()
        n = C.c_int()
        _lib.check(self._h, self._L.sdr_profile_read(self._h, f, b, 256, C.byref(n)))
        return list(f[:n.value]), list(b[:n.value])

    def export_frames(self, frames_tensor, counts_tensor):
        """Copy the last batch's frame records / per-capture counts into torch
        tensors on this GPU (uint8 [ncap*max_frames*96], int32 [ncap])."""
        _lib.check(self._h, self._L.sdr_export_frames(
            self._h, C.c_void_p(frames_tensor.data_ptr()), frames_tensor.numel() * frames_tensor.element_size(),
            C.c_void_p(counts_tensor.data_ptr()), counts_tensor.numel() * counts_tensor.element_size()))

    # -- timing on the stream the kernels run on
    def timer_start(self):
        _lib.check(self._h, self._L.sdr_timer_start(self._h))

    def timer_stop(self):
        _lib.check(self._h, self._L.sdr_timer_stop(self._h))

    def timer_elapsed_ms(self):
        ms = C.c_float()
        _lib.check(self._h, self._L.sdr_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value
