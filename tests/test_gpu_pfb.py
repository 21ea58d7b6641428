"""64-channel polyphase channelizer (BASELINE.json configs[4]) through the C ABI
against the oracle's orc_pfb_channelize (itself checked against the bank of 64
freq_xlating_fir_filter blocks in tests/test_oracle_dsp.py).  Bar: channel floats
bit-identical (stated tolerance 1e-5 relative, observed 0 differing words), sliced
bytes and frames bit-exact."""
import numpy as np
import pytest

from oracle import decoders as odec
from oracle import dsp as odsp
from sdr_amd import chain, synth
from tests.test_host_decode import assert_ts33c_equal

pytestmark = pytest.mark.gpu

FS = 20e6
M = 64


def _proto(fs=FS):
    return chain.low_pass(1.0, fs, 100e3, 100e3)   # 481 taps -> 8 per branch


@pytest.mark.parametrize("ntaps,n,ncap", [(481, 64 * 300 + 17, 1), (200, 64 * 129, 2), (1000, 64 * 64 * 5 + 63, 2),
                                          (481, 64 * 1000 + 1, 3), (64, 4096, 1), (481, 100, 1)])
def test_pfb_floats_bit_identical(ntaps, n, ncap):
    rng = np.random.default_rng(ntaps + n)
    x = (rng.standard_normal((ncap, n)) + 1j * rng.standard_normal((ncap, n))).astype(np.complex64)
    taps = _proto()[:ntaps] if ntaps <= 481 else rng.standard_normal(ntaps).astype(np.float32) / ntaps
    with chain.Chain(FS, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=ncap, max_samples=n,
                     channels=M) as c:
        c.batch_process(x)
        y = c.fetch_demod().reshape(ncap, M, -1)
    for i in range(ncap):
        want = odsp.pfb_channelize(x[i], taps, M)
        assert y[i].shape == want.shape == (M, n // M)
        assert np.array_equal(y[i].view(np.uint32), want.view(np.uint32)), i


def _slice_oracle(y, gain, lo, hi, keep):
    rows = []
    for c in range(y.shape[0]):
        b, _, _ = odsp.slice_bits(odsp.mag2(y[c]), gain, lo, hi, keep)
        rows.append(b)
    return np.stack(rows)


def test_pfb_wideband_emitters_decode_per_channel(kat):
    """Three emitters in three sub-bands of one 20 Msps capture, all decoders run
    over every channel: frames come out of the right rows, bit-exact vs the oracle."""
    n = 1 << 22
    keep = 16
    rate = FS / M / keep
    rng = np.random.default_rng(9)
    pkt = synth.ts33c_packet(channel=2, rolling_code=7, temp_tenths=-37, humidity=61, filler=(1, 2, 3, 4, 5))
    kw = [int(ch) for ch in kat["kwmobile_sniff"]["bits"]]
    em = [(5 * FS / M, synth.ts33c_symbols(synth.ts33c_bits(pkt), rng) * 2, 5000.0),
          ((40 - 64) * FS / M + 20e3, synth.kwmobile_symbols(kw, repeats=3), 4000.0),
          (17 * FS / M - 15e3, synth.mumbi_symbols(bytes(kat["mumbi_codes"]["b_off"]), repeats=4), 9000.0)]
    x = synth.wideband_capture(n, FS, em, seed=5)
    taps = _proto()
    y = odsp.pfb_channelize(x, taps, M)
    want = _slice_oracle(y, 1.0, 0.05, 0.15, keep)
    with chain.Chain(FS, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=n,
                     max_frames=64, channels=M) as c:
        out = c.batch_process(x[None, :])
        bits = c.fetch_bits().reshape(M, -1)
        assert out.n_bits == want.shape[1] == n // M // keep
        assert np.array_equal(bits, want)
        f_ts = c.fetch_frames()
        c.redecode(chain.DEC_KWMOBILE)
        f_kw = c.fetch_frames()
        c.redecode(chain.DEC_MUMBI)
        f_mu = c.fetch_frames()
        # all of them (and two more) in ONE launch: identical frame records per decoder
        import ctypes as C

        def rec(fr):
            return [bytes(bytearray(C.string_at(C.byref(f), C.sizeof(f)))) for f in fr]
        decs = [chain.DEC_TS33C, chain.DEC_KWMOBILE, chain.DEC_MUMBI, chain.DEC_SOMFY, chain.DEC_BERNER]
        c.redecode_multi(decs)
        multi = [c.fetch_frames_multi(k) for k in range(len(decs))]
        assert rec(multi[0]) == rec(f_ts) and rec(multi[1]) == rec(f_kw) and rec(multi[2]) == rec(f_mu)
        c.redecode(chain.DEC_BERNER)
        assert rec(multi[4]) == rec(c.fetch_frames())
        c.redecode(chain.DEC_SOMFY)
        assert rec(multi[3]) == rec(c.fetch_frames())
        c.batch_process(x[None, :])                  # the handle's own tables are untouched by the multi pass
        assert rec(c.fetch_frames()) == rec(f_ts)
    o = odec.TS33CStreamDecoder(sample_rate=rate)
    o.work([want[5]])
    good = [f for f in f_ts if f.status == 0]
    assert {f.capture for f in good} == {5} and len(good) == len(o.frames) == 2
    for of, gf in zip(o.frames, good):
        assert_ts33c_equal(of, gf)
        assert gf.packet() == pkt
    s = odec.KwmobileDecoder()
    for lvl, tick in odec.edges_to_events(want[40], int(rate), watchdog_us=3000):
        s.cbf(lvl, tick)
    assert {f.capture for f in f_kw} == {40}
    assert [f.packet() for f in f_kw] == [f["frame"] for f in s.frames] and len(f_kw) >= 1
    assert f_kw[0].packet().hex() == kat["kwmobile_sniff"]["frame_hex"]
    s = odec.MumbiSniffer()
    for lvl, tick in odec.edges_to_events(want[17], int(rate), watchdog_us=8000):
        s.cbf(lvl, tick)
    assert {f.capture for f in f_mu} == {17}
    assert [f.packet() for f in f_mu] == s.frames and len(f_mu) >= 1


def test_pfb_argument_checks():
    from sdr_amd import _lib
    with pytest.raises(_lib.SdrError):
        chain.Chain(FS, channels=32, max_samples=4096)
    with pytest.raises(_lib.SdrError):
        chain.Chain(FS, channels=64, demod=chain.DEMOD_QUAD, max_samples=4096)
    with chain.Chain(FS, channels=64, taps=_proto(), max_samples=4096, max_batch=2) as c:
        with pytest.raises(_lib.SdrError):       # only a one-capture channelizer streams
            c.process_iq(np.zeros(64, np.complex64))
    with chain.Chain(FS, channels=64, taps=_proto(), max_samples=4096) as c:
        with pytest.raises(_lib.SdrError):
            c.set_taps(np.ones(1025, np.float32))
        out = c.batch_process(np.zeros((1, 63), np.complex64))   # fewer samples than one output
        assert out.n_bits == 0
