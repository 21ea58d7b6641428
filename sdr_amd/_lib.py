"""ctypes binding of libsdrhip.so (include/sdrhip.h).  Thin: struct mirrors,
argtypes, error mapping.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C sdr_amd/csrc``; there is no fallback
if it is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDR_LIB_PATH") or os.path.join(_HERE, "libsdrhip.so")  # env override: tuning experiments only

SDR_OK, SDR_E_ARG, SDR_E_HIP, SDR_E_CAPACITY, SDR_E_OVERFLOW, SDR_E_NODEVICE = 0, -1, -2, -3, -4, -5

WIN_HAMMING, WIN_HANN, WIN_BLACKMAN, WIN_RECTANGULAR, WIN_KAISER, WIN_BLACKMAN_HARRIS = range(6)
DEMOD_NONE, DEMOD_MAG2, DEMOD_QUAD = 0, 1, 2
IN_FC32, IN_CU8 = 0, 1
DEC_NONE, DEC_TS33C, DEC_TFA, DEC_TFA_GRC, DEC_SOMFY, DEC_BERNER, DEC_MUMBI, DEC_KWMOBILE, DEC_BALDR, DEC_VELUX = range(10)

FRAME_OK = 0
FRAME_BAD_LENGTH = 1
FRAME_PARITY = 2
FRAME_UNKNOWN_HEADER = 3
FRAME_CRC = 4
FRAME_SYMBOL_LENGTH = 5
FRAME_CHECKSUM = 6
FRAME_EXC_RESHAPE = 16
FRAME_EXC_INDEX = 17
FRAME_EXC_NAN = 18
FRAME_EXC_SHORT = 19

#: the reference's debug strings for each status (TS33C/stream_decoder.py:198-202,
#: TFA/stream_decoder.py:197-199, TFA/analyzer2.grc:960)
STATUS_TEXT = {
    FRAME_OK: "ok",
    FRAME_BAD_LENGTH: "Invalid packet length",
    FRAME_PARITY: "Parity check failed",
    FRAME_UNKNOWN_HEADER: "Unknown Sensor or Header",
    FRAME_CRC: "CRC-Check failed",
    FRAME_SYMBOL_LENGTH: "Symbol length error",
    FRAME_CHECKSUM: "checksum error",
}


class SdrError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "libsdrhip error %d: %s" % (code, text))
        self.code = code


class ChainCfg(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_double),
        ("center_freq", C.c_double),
        ("decim", C.c_int32),
        ("demod", C.c_int32),
        ("demod_gain", C.c_float),
        ("thr_lo", C.c_float),
        ("thr_hi", C.c_float),
        ("thr_init", C.c_int32),
        ("keep_n", C.c_int32),
        ("decoder", C.c_int32),
        ("decoder_rate", C.c_double),
        ("post_ntaps", C.c_int32),
        ("tap_demod", C.c_int32),
        ("max_batch", C.c_int32),
        ("max_frames", C.c_int32),
        ("max_samples", C.c_uint64),
    ]


class Frame(C.Structure):
    _fields_ = [
        ("capture", C.c_int32),
        ("decoder", C.c_int32),
        ("status", C.c_int32),
        ("nbits", C.c_int32),
        ("nbytes", C.c_int32),
        ("bytes", C.c_uint8 * 16),
        ("raw", C.c_uint8 * 8),
        ("v", C.c_int32 * 4),
        ("temp_tenths", C.c_int32),
        ("pad_", C.c_int32),
        ("pad2_", C.c_int32),
        ("lim", C.c_double * 2),
        ("end_pos", C.c_int64),
    ]

    def packet(self):
        if self.nbytes > 16:   # velux: the packet runs on through raw and v (40 contiguous bytes)
            return C.string_at(C.addressof(self) + Frame.bytes.offset, min(self.nbytes, 40))
        return bytes(bytearray(self.bytes[: self.nbytes]))


class Out(C.Structure):
    _fields_ = [
        ("n_in", C.c_uint64),
        ("n_bits", C.c_uint64),
        ("n_frames", C.c_uint64),
        ("overflow", C.c_int32),
        ("pad_", C.c_int32),
    ]


class FmCfg(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_double),
        ("center_freq", C.c_double),
        ("resamp_decim", C.c_int32),
        ("audio_decim", C.c_int32),
        ("lpf_fs", C.c_double),
        ("lpf_fc", C.c_double),
        ("lpf_tw", C.c_double),
        ("quad_rate", C.c_double),
        ("tau", C.c_double),
        ("audio_interp", C.c_int32),
        ("audio_decimation", C.c_int32),
        ("volume", C.c_float),
        ("pad_", C.c_int32),
        ("max_samples", C.c_uint64),
    ]


class PreambleInfo(C.Structure):
    _fields_ = [
        ("state", C.c_int32),
        ("preamble", C.c_int32),
        ("position", C.c_int64),
        ("events", C.c_int32),
        ("sum8", C.c_int32),
        ("preamble_pos", C.c_int64),
    ]


class SegmentPieces(C.Structure):
    _fields_ = [
        ("n_bits", C.c_int64),
        ("keep_from", C.c_int64),
        ("keep_to", C.c_int64),
        ("n_head", C.c_int32),
        ("n_tail", C.c_int32),
        ("n_runs", C.c_int32),
        ("n_frames", C.c_int32),
        ("first_level", C.c_int32),
        ("last_level", C.c_int32),
        ("flags", C.c_int32),
        ("pad_", C.c_int32),
    ]


class SegmentInfo(C.Structure):
    _fields_ = [
        ("first_bit_index", C.c_uint64),
        ("n_bits", C.c_uint64),
        ("n_undetermined", C.c_uint64),
        ("has_event", C.c_int32),
        ("last_state", C.c_int32),
        ("carry_in", C.c_int32),
        ("pad_", C.c_int32),
    ]


#: every symbol include/sdrhip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "sdr_firdes_low_pass", "sdr_resampler_taps", "sdr_create", "sdr_destroy", "sdr_set_taps",
    "sdr_set_post_taps", "sdr_set_stream", "sdr_synchronize", "sdr_last_error", "sdr_process_iq",
    "sdr_process_bits", "sdr_batch_process", "sdr_fetch_frames", "sdr_fetch_bits", "sdr_fetch_demod",
    "sdr_decoder_state", "sdr_timer_start", "sdr_timer_stop", "sdr_timer_elapsed_ms", "sdr_version",
    "sdr_profile_enable", "sdr_profile_read", "sdr_export_frames", "sdr_set_squelch",
    "sdr_fetch_squelch_counts", "sdr_preamble_create", "sdr_preamble_destroy", "sdr_preamble_work",
    "sdr_preamble_last_error", "sdr_set_prefilter",
    "sdr_fm_create", "sdr_fm_destroy", "sdr_fm_set_stream", "sdr_fm_process", "sdr_fm_fetch",
    "sdr_fm_last_error", "sdr_fm_reset", "sdr_fm_set_overlap",
    "sdr_pfb_create", "sdr_redecode", "sdr_set_input_format", "sdr_history_len", "sdr_seek", "sdr_segment_info_get", "sdr_export_bits",
    "sdr_batch_process_static", "sdr_static_scratch_bytes", "sdr_segment_info_rows", "sdr_batch_process_bits",
    "sdr_redecode_multi", "sdr_fetch_frames_multi", "sdr_set_overlap", "sdr_result_stream", "sdr_overlap_state",
    "sdr_fetch_pending_symbols", "sdr_fetch_raw_buffer", "sdr_fetch_runs",
    "sdr_segment_decode", "sdr_process_runs",
]

_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's).  If
    libsdrhip pulled in the system copy first, a later `import torch` would load a
    second HIP runtime into the process and fail with "No HIP GPUs are available".
    Loading torch's copy first (by path, torch itself is not imported) makes both
    bind to one runtime, whichever is used first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libsdrhip.so; raises if it has not been built (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with __graft_entry__.build() "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    fp = C.POINTER(C.c_float)
    L.sdr_firdes_low_pass.restype = C.c_int
    L.sdr_firdes_low_pass.argtypes = [C.c_double] * 4 + [C.c_int, C.c_double, fp, C.c_int]
    L.sdr_resampler_taps.restype = C.c_int
    L.sdr_resampler_taps.argtypes = [C.c_int, C.c_int, C.c_double, fp, C.c_int]
    L.sdr_create.restype = C.c_int
    L.sdr_create.argtypes = [C.POINTER(ChainCfg), C.POINTER(H)]
    L.sdr_destroy.restype = C.c_int
    L.sdr_destroy.argtypes = [H]
    L.sdr_set_taps.restype = C.c_int
    L.sdr_set_taps.argtypes = [H, fp, C.c_int, C.c_int]
    L.sdr_set_post_taps.restype = C.c_int
    L.sdr_set_post_taps.argtypes = [H, fp, C.c_int]
    L.sdr_set_prefilter.restype = C.c_int
    L.sdr_set_prefilter.argtypes = [H, fp, C.c_int, C.c_int]
    L.sdr_set_squelch.restype = C.c_int
    L.sdr_set_squelch.argtypes = [H, C.c_double, C.c_double, C.c_int]
    L.sdr_fetch_squelch_counts.restype = C.c_int
    L.sdr_fetch_squelch_counts.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_set_stream.restype = C.c_int
    L.sdr_set_stream.argtypes = [H, C.c_void_p]
    L.sdr_synchronize.restype = C.c_int
    L.sdr_synchronize.argtypes = [H]
    L.sdr_set_overlap.restype = C.c_int
    L.sdr_set_overlap.argtypes = [H, C.c_int]
    L.sdr_result_stream.restype = C.c_void_p
    L.sdr_result_stream.argtypes = [H]
    L.sdr_overlap_state.restype = C.c_int
    L.sdr_overlap_state.argtypes = [H, C.POINTER(C.c_int32 * 8)]
    L.sdr_last_error.restype = C.c_char_p
    L.sdr_last_error.argtypes = [H]
    L.sdr_process_iq.restype = C.c_int
    L.sdr_process_iq.argtypes = [H, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Out)]
    L.sdr_process_bits.restype = C.c_int
    L.sdr_process_bits.argtypes = [H, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Out)]
    L.sdr_batch_process.restype = C.c_int
    L.sdr_batch_process.argtypes = [H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(Out)]
    L.sdr_fetch_frames.restype = C.c_int
    L.sdr_fetch_frames.argtypes = [H, C.POINTER(Frame), C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_fetch_bits.restype = C.c_int
    L.sdr_fetch_bits.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_fetch_demod.restype = C.c_int
    L.sdr_fetch_demod.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_fetch_pending_symbols.restype = C.c_int
    L.sdr_fetch_pending_symbols.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_fetch_raw_buffer.restype = C.c_int
    L.sdr_fetch_raw_buffer.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_fetch_runs.restype = C.c_int
    L.sdr_fetch_runs.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_decoder_state.restype = C.c_int
    L.sdr_decoder_state.argtypes = [H, C.POINTER(C.c_int32 * 4)]
    for name in ("sdr_timer_start", "sdr_timer_stop"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [H]
    L.sdr_timer_elapsed_ms.restype = C.c_int
    L.sdr_timer_elapsed_ms.argtypes = [H, C.POINTER(C.c_float)]
    L.sdr_version.restype = C.c_int
    L.sdr_profile_enable.restype = C.c_int
    L.sdr_profile_enable.argtypes = [H, C.c_int]
    L.sdr_profile_read.restype = C.c_int
    L.sdr_profile_read.argtypes = [H, fp, fp, C.c_int, C.POINTER(C.c_int)]
    L.sdr_export_frames.restype = C.c_int
    L.sdr_export_frames.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.sdr_pfb_create.restype = C.c_int
    L.sdr_pfb_create.argtypes = [C.POINTER(ChainCfg), C.c_int, C.POINTER(H)]
    L.sdr_redecode.restype = C.c_int
    L.sdr_redecode.argtypes = [H, C.c_int, C.c_double]
    L.sdr_redecode_multi.restype = C.c_int
    L.sdr_redecode_multi.argtypes = [H, C.POINTER(C.c_int32), C.c_int, C.c_double]
    L.sdr_fetch_frames_multi.restype = C.c_int
    L.sdr_fetch_frames_multi.argtypes = [H, C.c_int, C.POINTER(Frame), C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_set_input_format.restype = C.c_int
    L.sdr_set_input_format.argtypes = [H, C.c_int]
    L.sdr_history_len.restype = C.c_int
    L.sdr_history_len.argtypes = [H]
    L.sdr_seek.restype = C.c_int
    L.sdr_seek.argtypes = [H, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]
    L.sdr_segment_info_get.restype = C.c_int
    L.sdr_segment_info_get.argtypes = [H, C.POINTER(SegmentInfo)]
    L.sdr_segment_info_rows.restype = C.c_int
    L.sdr_segment_info_rows.argtypes = [H, C.POINTER(SegmentInfo), C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_batch_process_bits.restype = C.c_int
    L.sdr_batch_process_bits.argtypes = [H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(Out)]
    L.sdr_segment_decode.restype = C.c_int
    L.sdr_segment_decode.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(SegmentPieces)]
    L.sdr_process_runs.restype = C.c_int
    L.sdr_process_runs.argtypes = [H, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Out)]
    L.sdr_export_bits.restype = C.c_int
    L.sdr_export_bits.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_batch_process_static.restype = C.c_int
    L.sdr_batch_process_static.argtypes = [H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Out)]
    L.sdr_static_scratch_bytes.restype = C.c_size_t
    L.sdr_static_scratch_bytes.argtypes = []
    L.sdr_fm_create.restype = C.c_int
    L.sdr_fm_create.argtypes = [C.POINTER(FmCfg), C.POINTER(H)]
    L.sdr_fm_destroy.restype = C.c_int
    L.sdr_fm_destroy.argtypes = [H]
    L.sdr_fm_set_stream.restype = C.c_int
    L.sdr_fm_set_stream.argtypes = [H, C.c_void_p]
    L.sdr_fm_process.restype = C.c_int
    L.sdr_fm_process.argtypes = [H, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]
    L.sdr_fm_set_overlap.restype = C.c_int
    L.sdr_fm_set_overlap.argtypes = [H, C.c_int]
    L.sdr_fm_reset.restype = C.c_int
    L.sdr_fm_reset.argtypes = [H]
    L.sdr_fm_fetch.restype = C.c_int
    L.sdr_fm_fetch.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdr_fm_last_error.restype = C.c_char_p
    L.sdr_fm_last_error.argtypes = [H]
    L.sdr_preamble_create.restype = C.c_int
    L.sdr_preamble_create.argtypes = [C.POINTER(H)]
    L.sdr_preamble_destroy.restype = C.c_int
    L.sdr_preamble_destroy.argtypes = [H]
    L.sdr_preamble_work.restype = C.c_int
    L.sdr_preamble_work.argtypes = [H, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(PreambleInfo)]
    L.sdr_preamble_last_error.restype = C.c_char_p
    L.sdr_preamble_last_error.argtypes = [H]
    # host-compiled test hooks (csrc/selftest.h); not used by the product path
    L.sdr_selftest_hist_mid_edge.restype = C.c_int
    L.sdr_selftest_hist_mid_edge.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.sdr_selftest_pairwise_mean.restype = C.c_int
    L.sdr_selftest_pairwise_mean.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.sdr_selftest_dec_create.restype = C.c_void_p
    L.sdr_selftest_dec_create.argtypes = [C.c_int, C.c_double]
    L.sdr_selftest_dec_destroy.restype = None
    L.sdr_selftest_dec_destroy.argtypes = [C.c_void_p]
    L.sdr_selftest_dec_work.restype = C.c_int
    L.sdr_selftest_dec_work.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Frame), C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_int32 * 4)]
    _lib = L
    return L


def check(handle, rc):
    if rc != SDR_OK:
        L = load()
        text = L.sdr_last_error(handle)
        raise SdrError(rc, text.decode("utf-8", "replace") if text else "")
