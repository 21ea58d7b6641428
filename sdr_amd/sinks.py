"""GRC-compatible sinks so that the outputs of a chain can be
diffed against a GNU Radio run made elsewhere.  File formats only -- no signal
processing happens here; the arrays come from Chain.fetch_bits()/fetch_demod().

  file_sink     blocks_file_sink: raw native-endian items appended to a file
                (ref: TS33C/decoder.grc:507-556 `type byte`, the .dat that
                TS33C/stream_decoder.py:226 reads back with np.fromfile(int8);
                complex items = interleaved float32 pairs, the `.cfile` format).
  wavfile_sink  blocks_wavfile_sink [GR-ext: gr-blocks/lib/wavfile_sink_impl.cc
                3.7.11, unpinned]: PCM WAV, one float stream per channel,
                interleaved; 8 bit = unsigned, iround((x + 1) * 127) clipped to
                0..255; 16 bit = iround(x * 32767) clipped to +-32767
                (ref: TS33C/decoder.grc:819-913, bits_per_sample 8, nchan 1 / 4).
"""
import struct

import numpy as np


class file_sink(object):
    def __init__(self, path, dtype, append=False):
        self.dtype = np.dtype(dtype)
        self._f = open(path, "ab" if append else "wb")

    def work(self, items):
        a = np.ascontiguousarray(items, self.dtype)
        a.tofile(self._f)
        return a.size

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _iround(x):
    """boost::math::iround: nearest, halves away from zero."""
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))


class wavfile_sink(object):
    def __init__(self, path, n_channels, sample_rate, bits_per_sample=16):
        if bits_per_sample not in (8, 16):
            raise ValueError("bits_per_sample must be 8 or 16")
        self.nchan, self.rate, self.bits = int(n_channels), int(sample_rate), int(bits_per_sample)
        self._f = open(path, "wb")
        self._bytes = 0
        self._write_header()

    def _write_header(self):
        bps = self.bits // 8
        self._f.seek(0)
        self._f.write(b"RIFF" + struct.pack("<I", 36 + self._bytes) + b"WAVE")
        self._f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, self.nchan, self.rate, self.rate * self.nchan * bps,
                                            self.nchan * bps, self.bits))
        self._f.write(b"data" + struct.pack("<I", self._bytes))

    def work(self, input_items):
        """input_items: one float array per channel, equal lengths."""
        if len(input_items) != self.nchan:
            raise ValueError("expected %d channels" % self.nchan)
        x = np.stack([np.asarray(c, np.float32) for c in input_items], axis=1).astype(np.float64)
        if self.bits == 8:
            q = np.clip(_iround((x + 1.0) * 127.0), 0, 255).astype(np.uint8)
        else:
            q = np.clip(_iround(x * 32767.0), -32767, 32767).astype("<i2")
        self._f.seek(0, 2)
        q.tofile(self._f)
        self._bytes += q.nbytes
        return x.shape[0]

    def close(self):
        self._write_header()   # blocks_wavfile_sink patches the sizes on close
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
