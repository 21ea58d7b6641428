"""Streaming state of the callback-rule decoders on the GPU: the reference's sniffers are continuous
(somfy/sniff.py:28-117, berner/sniff.py:21-44, mumbi/sniff.py:26-65, kwmobile/decoder.py:95-135,
baldr/decoder.py:87-142), so a frame cut by a sdr_process_bits / sdr_process_iq boundary must decode as if
the stream had come in one piece.  Oracle: the sniffer restatements fed edges_to_events of the WHOLE stream."""
import numpy as np
import pytest

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import chain, synth
from tests.test_host_decode import RULE_SET, _events_decode, _rule_stream, oracle_frame_key, rule_frame_key

pytestmark = pytest.mark.gpu


def _cuts(rng, n, k):
    c = sorted(set([0, n] + [int(v) for v in rng.integers(0, n, k)]))
    return c[:2] + [c[1]] + c[2:]      # one empty call


@pytest.mark.parametrize("seed", range(4))
def test_rules_from_bits_in_ragged_chunks(seed):
    rng = np.random.default_rng(1300 + seed)
    nframes = 0
    for kind, _, _, _ in RULE_SET:
        rate = int(rng.choice([100000, 250000, 1000000]))
        env = _rule_stream(rng, kind, rate)
        for _, decoder, cls, wd in RULE_SET:
            want = [oracle_frame_key(decoder, a) for a in _events_decode(cls, env, rate, wd)]
            for k in (5, 40):
                cuts = _cuts(rng, env.size, k)
                got = []
                with chain.Chain(rate, decoder=decoder, decoder_rate=rate, max_samples=env.size, max_frames=64) as c:
                    for a, b in zip(cuts[:-1], cuts[1:]):
                        c.process_bits(env[a:b])
                        got += [rule_frame_key(decoder, f) for f in c.fetch_frames()]
                assert got == want, (kind, decoder, k, rate)
            nframes += len(want)
    assert nframes >= 5


def test_rules_from_iq_chunks_and_restart():
    """the whole chain in streaming mode (filter history, threshold carry AND the rule's state cross the call
    boundaries); sdr_seek / set_taps restart the rule with the rest of the stream state"""
    rng = np.random.default_rng(77)
    fs, D = 1e6, 10
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    for kind, decoder, cls, wd in RULE_SET[:3]:
        env = _rule_stream(rng, kind, int(fs)).astype(np.uint8)
        iq = synth.ook_iq(env, fs, f_off=50e3, amp=0.7, sigma=0.02, rng=rng)
        n = iq.size - iq.size % D
        iq = iq[:n]
        wb = odsp.ook_chain(iq, taps, D, 1.0, 0.05, 0.15, 1, fc=50e3, fs=fs)
        want = [oracle_frame_key(decoder, a) for a in _events_decode(cls, wb, int(fs / D), wd)]
        assert want
        cuts = _cuts(rng, n, 9)
        with chain.Chain(fs, decim=D, center_freq=50e3, taps=taps, decoder=decoder, max_samples=n, max_frames=64) as c:
            for rep in range(2):
                got, bits = [], []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    c.process_iq(iq[a:b])
                    bits.append(c.fetch_bits())
                    got += [rule_frame_key(decoder, f) for f in c.fetch_frames()]
                assert np.array_equal(np.concatenate(bits), wb)
                assert got == want, (kind, rep)
                c.seek(0)              # back to stream start: the second pass must not see the first one's state
