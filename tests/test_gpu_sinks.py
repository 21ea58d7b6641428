"""SURVEY 8f-3 end to end on the GPU: chain output -> file_sink (the `.dat` byte file of TS33C/decoder.grc:507-556)
-> read back exactly as the reference's main() does (TS33C/stream_decoder.py:222-235: np.fromfile(int8),
np.array_split(raw, 100), decoder.work([chunk]) per piece) -> frames equal the oracle decoder fed the same file;
and the 4-channel wavfile_sink debug tap (TS33C/decoder.grc:819-913) of the demodulated floats."""
import struct

import numpy as np
import pytest

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import chain, sinks, synth, ts33c, tfa

pytestmark = pytest.mark.gpu


def test_ts33c_chain_to_file_sink_to_stream_decoder(tmp_path, capsys):
    fs, D, n = 1e6, 50, 1 << 20
    iq, pkt, _ = synth.ts33c_capture(n, fs, seed=5, f_off=100e3, n_frames=9)
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    path = str(tmp_path / "bits.dat")
    # flowgraph: the chain's sliced stream goes to a file_sink in ragged streaming calls (appending)
    with chain.Chain(fs, decim=D, center_freq=100e3, taps=taps, max_samples=n) as c, sinks.file_sink(path, np.int8) as sink:
        for a, b in zip([0, 1000, 300001, 300002, 700000], [1000, 300001, 300002, 700000, n]):
            c.process_iq(iq[a:b])
            sink.work(c.fetch_bits())
    # the reference's main(): fromfile -> array_split(100) -> work([chunk])
    raw_data = np.ravel(np.fromfile(path, dtype=np.int8))
    assert np.array_equal(raw_data, odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, 1, fc=100e3, fs=fs))
    got = ts33c.StreamDecoder(sample_rate=20000)
    want = odec.TS33CStreamDecoder(sample_rate=20000)
    for chunk in np.array_split(raw_data, 100):
        got.work([chunk])
        want.work([chunk])
        assert (got.temperature, got.humidity) == (want.temperature, want.humidity)
        assert (got.state, got.chunk_id, got.previousSymbolType) == (want.state, want.chunk_id, want.previousSymbolType)
    assert [f["packet"] for f in got.frames] == [f["packet"] for f in want.frames] and len(got.frames) >= 6
    assert [f["status"] for f in got.frames] == [f["status"] for f in want.frames]
    assert all(f["packet"] == pkt for f in got.frames)
    got.close()
    capsys.readouterr()


def test_tfa_chain_to_file_sink_200_pieces(tmp_path, capsys):
    """TFA/stream_decoder.py:220-237: the same harness with 200 pieces at 2 MS/s (analyzer2 chain: mag^2 -> fff LPF)"""
    fs = 2e6
    rng = np.random.default_rng(8)
    sym, pkts = [], []
    for _ in range(5):
        p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10, humidity=int(rng.integers(0, 100)))
        pkts.append(p)
        sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=3000.0)
    env = synth.envelope(sym, fs, lead_in_us=900.0)
    env = np.concatenate([env, np.zeros(4000, np.uint8)])
    iq = synth.ook_iq(env, fs, f_off=0.0, amp=0.9, sigma=0.01, rng=rng)
    post = chain.low_pass(1.0, fs, 500e3, 1e6, chain.WIN_BLACKMAN)      # 7 taps, TFA/analyzer2.grc:1075-1147
    path = str(tmp_path / "tfa_bits.dat")
    with chain.Chain(fs, decim=1, post_taps=post, thr=(0.2, 0.4), max_samples=iq.size) as c, sinks.file_sink(path, np.int8) as sink:
        c.process_iq(iq)
        sink.work(c.fetch_bits())
    raw_data = np.ravel(np.fromfile(path, dtype=np.int8))
    got = tfa.StreamDecoder(sample_rate=int(fs))
    want = odec.TFAStreamDecoder(sample_rate=int(fs))
    for chunk in np.array_split(raw_data, 200):
        got.work([chunk])
        want.work([chunk])
    assert [f["packet"] for f in got.frames] == [f["packet"] for f in want.frames] == pkts
    assert (got.identifier, got.temperature, got.humidity, got.battery_ok) == (want.identifier, want.temperature, want.humidity, want.battery_ok)
    got.close()
    capsys.readouterr()


def test_wavfile_sink_debug_tap_of_demod_floats(tmp_path):
    """TS33C/decoder.grc:819-913: 8-bit wavfile_sink on the float streams; the chain's demodulated floats written
    through the sink read back as iround((x+1)*127) of the oracle's floats"""
    fs, D, n = 1e6, 50, 1 << 17
    iq, _, _ = synth.ts33c_capture(n, fs, seed=6, f_off=0.0, n_frames=1)
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    with chain.Chain(fs, decim=D, taps=taps, tap_demod=True, max_samples=n) as c:
        c.process_iq(iq)
        v = c.fetch_demod()
        b = c.fetch_bits().astype(np.float32)
    want_v = odsp.mag2(odsp.fir_ccf(iq, taps, D))
    assert np.array_equal(v.view(np.uint32), want_v.view(np.uint32))
    path = str(tmp_path / "tap.wav")
    with sinks.wavfile_sink(path, 2, int(fs / D), bits_per_sample=8) as w:
        w.work([v, b])
    blob = open(path, "rb").read()
    assert blob[:4] == b"RIFF" and struct.unpack("<I", blob[40:44])[0] == 2 * v.size
    pcm = np.frombuffer(blob[44:], np.uint8).reshape(-1, 2)
    q = np.clip(np.where(want_v.astype(np.float64) + 1.0 >= 0, np.floor((want_v.astype(np.float64) + 1.0) * 127.0 + 0.5), 0), 0, 255)
    assert np.array_equal(pcm[:, 0], q.astype(np.uint8)) and set(np.unique(pcm[:, 1])) <= {127, 254}
