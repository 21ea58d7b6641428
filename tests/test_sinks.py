"""GRC-compatible sink formats (host side, no GPU): the byte .dat round-trips
through np.fromfile as TS33C/stream_decoder.py:226 reads it; WAV files read back
with scipy and hold GNU Radio's quantisation."""
import numpy as np
import scipy.io.wavfile as wav

from sdr_amd import sinks


def test_file_sink_formats(tmp_path):
    bits = (np.arange(1000) % 3 == 0).astype(np.int8)
    p = tmp_path / "bits.dat"
    with sinks.file_sink(str(p), np.int8) as s:
        s.work(bits[:400])
        s.work(bits[400:])
    assert np.array_equal(np.ravel(np.fromfile(str(p), dtype=np.int8)), bits)
    z = (np.arange(50) + 1j * np.arange(50)[::-1]).astype(np.complex64)
    p = tmp_path / "iq.cfile"
    with sinks.file_sink(str(p), np.complex64) as s:
        s.work(z)
    assert np.array_equal(np.fromfile(str(p), np.float32).reshape(-1, 2), z.view(np.float32).reshape(-1, 2))
    with sinks.file_sink(str(p), np.complex64, append=True) as s:
        s.work(z[:5])
    assert np.fromfile(str(p), np.complex64).size == 55


def test_wavfile_sink_8bit_4ch_and_16bit(tmp_path):
    t = np.linspace(-1.2, 1.2, 97).astype(np.float32)
    chans = [t, -t, 0.5 * t, np.zeros_like(t)]
    p = tmp_path / "taps.wav"
    with sinks.wavfile_sink(str(p), 4, 20000, bits_per_sample=8) as s:
        s.work([c[:40] for c in chans])
        s.work([c[40:] for c in chans])
    rate, data = wav.read(str(p))
    assert rate == 20000 and data.shape == (97, 4) and data.dtype == np.uint8
    assert data[0, 0] == 0 and data[-1, 0] == 255 and data[48, 0] == 127 and np.all(data[:, 3] == 127)
    assert data[60, 2] == int(np.floor((0.5 * float(t[60]) + 1.0) * 127.0 + 0.5))
    p = tmp_path / "one.wav"
    with sinks.wavfile_sink(str(p), 1, 48000) as s:
        s.work([np.array([0.0, 1.0, -1.0, 0.25, 2.0], np.float32)])
    rate, data = wav.read(str(p))
    assert rate == 48000 and data.tolist() == [0, 32767, -32767, 8192, 32767]
