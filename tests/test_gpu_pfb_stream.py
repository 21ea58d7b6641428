"""The 64-channel channelizer as a STREAM and in SEGMENTS (BASELINE.json configs[4]: "20 Msps capture
channelized into 64 sub-bands, all sensor decoders per channel on 8 GPUs"): ragged sdr_process_iq chunks ==
one sdr_batch_process pass == the oracle on all 64 rows (floats bit-identical, sliced bytes and frames
bit-exact); contiguous segments positioned with sdr_seek + the (P-1)*64-sample halo, stitched with the per-row
threshold summaries, equal the single pass."""
import numpy as np
import pytest

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import chain, segments, synth
from tests.test_gpu_pfb import FS, M, _proto, _slice_oracle
from tests.test_host_decode import assert_ts33c_equal

pytestmark = pytest.mark.gpu


def _capture(n, keep, seed=21):
    rng = np.random.default_rng(seed)
    pkt = synth.ts33c_packet(channel=3, rolling_code=5, temp_tenths=123, humidity=44, filler=(9, 8, 7, 6, 5))
    codes = sorted(synth.MUMBI_CODES)
    code = bytes(bytearray(synth.MUMBI_CODES[codes[3]]))
    em = [(7 * FS / M, synth.ts33c_symbols(synth.ts33c_bits(pkt), rng) * 2, 5000.0),
          ((50 - 64) * FS / M + 10e3, synth.mumbi_symbols(code, repeats=5), 9000.0),
          (21 * FS / M, synth.berner_symbols(0x02EAABBF, repeats=5), 60000.0)]
    return synth.wideband_capture(n, FS, em, seed=seed), pkt, code


def test_pfb_streaming_floats_any_chunking():
    """complex channel outputs, chunk boundaries anywhere (inside a 64-sample output period, shorter than the
    filter, empty): bit-identical to the one-shot channelizer and the oracle"""
    rng = np.random.default_rng(3)
    n = 64 * 700 + 37
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    taps = _proto()
    want = odsp.pfb_channelize(x, taps, M)
    cuts = [0, 1, 63, 64, 65, 700, 700, 5000, 5001, 20000, 33333, n]
    with chain.Chain(FS, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_samples=n, channels=M) as c:
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            c.process_iq(x[a:b])
            parts.append(c.fetch_demod().reshape(M, -1))
        y = np.concatenate(parts, axis=1)
    assert y.shape == want.shape
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("keep", [16, 1])
def test_pfb_streaming_bits_and_frames(keep):
    n = 1 << 22 if keep == 16 else 1 << 19
    rate = FS / M / keep
    x, pkt, code = _capture(n, keep)
    taps = _proto()
    want = _slice_oracle(odsp.pfb_channelize(x, taps, M), 1.0, 0.05, 0.15, keep)
    rng = np.random.default_rng(keep)
    cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n, 9)]))
    res = {}
    for dec in (chain.DEC_TS33C, chain.DEC_MUMBI):
        with chain.Chain(FS, taps=taps, keep_n=keep, decoder=dec, decoder_rate=rate, max_samples=n, max_frames=64,
                         channels=M) as c:
            bits, frames = [], []
            for a, b in zip(cuts[:-1], cuts[1:]):
                c.process_iq(x[a:b])
                bits.append(c.fetch_bits().reshape(M, -1))
                frames += [(f.capture, f.status, f.packet()) for f in c.fetch_frames()]
            assert np.array_equal(np.concatenate(bits, axis=1), want)
            res[dec] = frames
    if keep == 16:
        o = odec.TS33CStreamDecoder(sample_rate=rate)
        o.work([want[7]])
        good = [f for f in res[chain.DEC_TS33C] if f[1] == 0]
        assert [f[2] for f in good] == [f["packet"] for f in o.frames if f["status"] == "ok"] == [pkt, pkt]
        assert {f[0] for f in good} == {7}
        s = odec.MumbiSniffer()
        for lvl, tick in odec.edges_to_events(want[50], int(rate), watchdog_us=8000):
            s.cbf(lvl, tick)
        assert [f[2] for f in res[chain.DEC_MUMBI] if f[0] == 50] == s.frames and len(s.frames) >= 3


def test_pfb_segments_stitched_equal_single_pass():
    """4 contiguous segments (boundaries multiples of 64*keep), each from sdr_seek with its halo and carry-in 0;
    stitch_rows resolves the per-channel threshold carries; the stitched rows equal the oracle's single pass
    and decode (sdr_batch_process_bits, all 64 rows at once) to the same frames as one batch pass."""
    keep, n = 16, 1 << 22
    rate = FS / M / keep
    x, pkt, code = _capture(n, keep, seed=22)
    # a stretch between the thresholds across the second boundary in channel 7, entered from a pulse
    taps = _proto()
    want = _slice_oracle(odsp.pfb_channelize(x, taps, M), 1.0, 0.05, 0.15, keep)
    bounds = segments.segment_bounds(n, 4, M * keep)
    bits_list, infos_list = [], []
    with chain.Chain(FS, taps=taps, keep_n=keep, decoder=chain.DEC_NONE, max_samples=n, channels=M) as c:
        H = c.history_len()
        assert H >= 7 * 64 + 63
        for a, b in bounds:
            lo = max(0, a - H)
            c.seek(a, x[lo:a] if a > lo else None)
            c.process_iq(x[a:b])
            infos_list.append(c.segment_info_rows())
            bits_list.append(c.fetch_bits().reshape(M, -1))
    assert all(len(i) == M for i in infos_list)
    stream = segments.stitch_rows(bits_list, infos_list, 0)
    assert np.array_equal(stream, want)
    # thresholds that leave the idle channel BETWEEN them (lo below the noise floor): a segment that starts idle
    # is undetermined up to its first pulse and takes the state the previous segments left
    want2 = _slice_oracle(odsp.pfb_channelize(x, taps, M), 1.0, 1e-12, 0.05, keep)
    bits_list, infos_list = [], []
    with chain.Chain(FS, taps=taps, keep_n=keep, thr=(1e-12, 0.05), decoder=chain.DEC_NONE, max_samples=n, channels=M) as c:
        for a, b in bounds:
            lo = max(0, a - H)
            c.seek(a, x[lo:a] if a > lo else None)
            c.process_iq(x[a:b])
            infos_list.append(c.segment_info_rows())
            bits_list.append(c.fetch_bits().reshape(M, -1))
    assert np.array_equal(segments.stitch_rows(bits_list, infos_list, 0), want2)
    assert any(inf[r].n_undetermined > 0 and want2[r].any() for inf in infos_list[1:] for r in (7, 21, 50))
    with chain.Chain(FS, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=n, max_frames=64,
                     channels=M) as c:
        c.batch_process(x[None, :])
        one = [(f.capture, f.status, f.packet(), f.end_pos) for f in c.fetch_frames()]
        c.batch_process_bits(stream)
        two = [(f.capture, f.status, f.packet(), f.end_pos) for f in c.fetch_frames()]
    assert one == two and [f[2] for f in one if f[1] == 0] == [pkt, pkt]


def test_pfb_streaming_short_chunk_then_max_samples_chunk():
    """A chunk shorter than the stream decoder's reset limit leaves its bytes pending in every channel row's
    rawBuffer (TS33C/stream_decoder.py:44-47); a max_samples chunk behind it must still fit -- the header promises
    chunks of any length.  max_samples is chosen so that the row stride had 2 bytes of slack before round 3
    (the pending margin was dropped for every handle with more than one row)."""
    keep = 16
    n_max = (1 << 22) - 2048
    rate = FS / M / keep
    short = 25 * keep * M                       # 25 bytes per row: 1280 us < reset_limit 1550 us
    x, pkt, code = _capture(short + n_max, keep, seed=23)
    taps = _proto()
    want = _slice_oracle(odsp.pfb_channelize(x, taps, M), 1.0, 0.05, 0.15, keep)
    with chain.Chain(FS, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=n_max, max_frames=64,
                     channels=M) as c:
        c.process_iq(x[:short])
        b1 = c.fetch_bits().reshape(M, -1)
        assert b1.shape[1] == 25 and not c.fetch_frames()
        c.process_iq(x[short:])
        b2 = c.fetch_bits().reshape(M, -1)
        frames = [(f.capture, f.status, f.packet()) for f in c.fetch_frames()]
    assert np.array_equal(np.concatenate([b1, b2], axis=1), want)
    o = odec.TS33CStreamDecoder(sample_rate=rate)
    o.work([want[7][:25]])
    o.work([want[7][25:]])
    assert [f[2] for f in frames if f[1] == 0 and f[0] == 7] == [f["packet"] for f in o.frames if f["status"] == "ok"] == [pkt, pkt]
