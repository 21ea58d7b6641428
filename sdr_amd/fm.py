"""FM-Radio flowgraph (FM-Radio/FM-Radio.grc:2903-2964) as one object on the
GPU path: xlating -> resampler 1/D -> low-pass -> wfm_rcv -> resampler 48/50 ->
volume.  Parameter names follow the GRC variables."""
import ctypes as C

import numpy as np

from . import _lib


class FmRadio(object):
    def __init__(self, samp_rate=2e6, shift=200e3, volume=1.0, lpf=(2e6, 400e3, 200e3), quad_rate=500e3,
                 audio_decimation=10, tau=75e-6, audio_resample=(48, 50), max_samples=1 << 24):
        self._L = _lib.load()
        self._h = C.c_void_p()
        cfg = _lib.FmCfg()
        cfg.sample_rate = float(samp_rate)
        cfg.center_freq = -float(shift)              # freq_xlating center_freq = -shift
        cfg.resamp_decim = int(samp_rate / 500e3)    # rational_resampler decimation int(samp_rate/500e3)
        cfg.audio_decim = int(audio_decimation)
        cfg.lpf_fs, cfg.lpf_fc, cfg.lpf_tw = [float(v) for v in lpf]
        cfg.quad_rate = float(quad_rate)
        cfg.tau = float(tau)
        cfg.audio_interp, cfg.audio_decimation = int(audio_resample[0]), int(audio_resample[1])
        cfg.volume = float(volume)
        cfg.max_samples = int(max_samples)
        self.cfg = cfg
        rc = self._L.sdr_fm_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise _lib.SdrError(rc, (self._L.sdr_fm_last_error(None) or b"").decode())

    def _check(self, rc):
        if rc != 0:
            raise _lib.SdrError(rc, (self._L.sdr_fm_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._L.sdr_fm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, ptr):
        self._check(self._L.sdr_fm_set_stream(self._h, C.c_void_p(ptr)))

    def set_overlap(self, on=True):
        """sdr_fm_set_overlap: the audio half of each process() call on a second stream beside the next call's
        channel filter (same results; fetch() waits for both streams)."""
        self._check(self._L.sdr_fm_set_overlap(self._h, int(bool(on))))

    def synchronize(self):
        """wait for everything process() has queued (both streams)"""
        n = C.c_size_t()
        self._check(self._L.sdr_fm_fetch(self._h, 4, None, 0, C.byref(n)))

    def reset(self):
        """Back to stream start (every block's state as after construction)."""
        self._check(self._L.sdr_fm_reset(self._h))

    def process(self, iq):
        """One scheduler pass over the next chunk of the stream (all block state persists between calls; the
        first call after construction / reset() is the capture from stream start); returns the audio sample
        count of this call."""
        if type(iq).__module__.startswith("torch") and iq.is_cuda:
            n = iq.numel() if iq.is_complex() else iq.numel() // 2
            ptr, dev, self._keep = iq.data_ptr(), 1, iq
        else:
            a = np.ascontiguousarray(iq, np.complex64)
            n, ptr, dev, self._keep = a.size, a.ctypes.data, 0, a
        na = C.c_size_t()
        self._check(self._L.sdr_fm_process(self._h, C.c_void_p(ptr), n, dev, C.byref(na)))
        return na.value

    def fetch(self, stage=4):
        """0 resampled complex, 1 discriminator, 2 audio FIR, 3 de-emphasis, 4 audio."""
        n = C.c_size_t()
        self._check(self._L.sdr_fm_fetch(self._h, stage, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            self._check(self._L.sdr_fm_fetch(self._h, stage, C.c_void_p(out.ctypes.data), out.size, C.byref(n)))
        return out.view(np.complex64) if stage == 0 else out
