"""Row a7: analog_pwr_squelch_cc between channel filter and demodulator
(TFA/decoder.grc:391-446 wiring: low_pass_filter -> pwr_squelch(gate) ->
quadrature_demod / complex_to_mag_squared), GPU vs oracle through the C ABI.

Pass/mute decisions and item counts must be identical; gated items are copies,
so demod floats and sliced bits are compared bit for bit."""
import numpy as np
import pytest

from oracle import dsp as odsp

pytestmark = pytest.mark.gpu


def _bursts(n, rng, fs=2e6, dev=60e3, bursts=((0.15, 0.35), (0.55, 0.8)), amp=0.7, noise=0.004):
    """FSK bursts over a weak noise floor: the squelch opens only inside the bursts."""
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * noise
    sym = np.repeat(rng.integers(0, 2, n // 52 + 1) * 2 - 1, 52)[:n]
    ph = 2 * np.pi * np.cumsum(sym * dev / fs)
    env = np.zeros(n)
    for a, b in bursts:
        env[int(a * n):int(b * n)] = amp
    return (x + env * np.exp(1j * ph)).astype(np.complex64)


def _oracle_stream(x, taps, db, alpha, gate, demod, gain, lo, hi, keep_n, chunks):
    """Chunked oracle with every block's state carried, as GNU Radio would."""
    H = taps.size - 1
    hist = np.zeros(H, np.complex64)
    sq = np.zeros(1, np.float64)
    prev = 0j
    thr_state, ctr = 0, keep_n
    floats, bits, counts = [], [], []
    for c in chunks:
        y = odsp.fir_ccf(c, taps, 1, hist=hist)
        hist = np.concatenate([hist, c])[-H:] if H else hist
        z = odsp.pwr_squelch(y, db, alpha, gate, state=sq)
        counts.append(z.size)
        if demod == "quad":
            v = odsp.quad_demod(z, 1.0, prev)
            if z.size:
                prev = complex(z[-1])
        else:
            v = odsp.mag2(z)
        floats.append(np.float32(gain) * v)
        b, thr_state, ctr = odsp.slice_bits(v, gain, lo, hi, keep_n, thr_state, ctr)
        bits.append(b)
    return floats, bits, counts


@pytest.mark.parametrize("demod", ["quad", "mag2"])
def test_squelch_streaming_ragged_chunks(demod):
    from sdr_amd import chain
    rng = np.random.default_rng(30)
    n = 60000
    x = _bursts(n, rng)
    taps = chain.low_pass(1.0, 2e6, 500e3, 250e3, chain.WIN_BLACKMAN)
    np.testing.assert_array_equal(taps, odsp.low_pass(1.0, 2e6, 500e3, 250e3, odsp.WIN_BLACKMAN))
    db, alpha = -20.0, 1e-3
    gain = 2e6 / (2 * np.pi * 300e3 / 8.0) if demod == "quad" else 1.0
    lo, hi = (-0.2, 0.2) if demod == "quad" else (0.1, 0.4)
    cuts = [0, 1, 1, 700, 5000, 5003, 20000, 20000, 41111, n]
    chunks = [x[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    want_f, want_b, want_c = _oracle_stream(x, taps, db, alpha, True, demod, gain, lo, hi, 3, chunks)
    assert 0 < sum(want_c) < n  # the gate both opens and closes
    c = chain.Chain(2e6, decim=1, taps=taps, demod=chain.DEMOD_QUAD if demod == "quad" else chain.DEMOD_MAG2,
                    demod_gain=gain, thr=(lo, hi), keep_n=3, tap_demod=True, max_samples=n,
                    squelch=(db, alpha, True))
    for i, ch in enumerate(chunks):
        c.process_iq(ch)
        assert int(c.fetch_squelch_counts()[0]) == want_c[i], i
        got_f = c.fetch_demod()
        assert got_f.size == want_c[i]
        assert np.array_equal(got_f.view(np.uint32), want_f[i].view(np.uint32)), i
        np.testing.assert_array_equal(c.fetch_bits(), want_b[i], err_msg="chunk %d" % i)
    c.close()


def test_squelch_batch_ragged_counts_and_ts33c_decode():
    """Batch behind a gating squelch: every capture keeps its own length on the
    device; OOK frames still decode from the gated stream (the gaps inside a frame
    survive because the power estimate decays slowly)."""
    from sdr_amd import chain, synth
    from oracle import decoders as odec
    rng = np.random.default_rng(7)
    fs, D = 1e6, 50
    n = 1 << 17
    caps = []
    for i in range(4):
        x, _, _ = synth.ts33c_capture(n, fs=fs, seed=100 + i, n_frames=1)
        # noise-only lead-in of a different length in front of every capture's frame
        k = 9000 * i
        lead = ((rng.standard_normal(k) + 1j * rng.standard_normal(k)) * 0.02).astype(np.complex64)
        caps.append(np.concatenate([lead, x])[:n])
    iq = np.stack(caps)
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    db, alpha = -25.0, 0.05
    c = chain.Chain(fs, decim=D, taps=taps, demod=chain.DEMOD_MAG2, thr=(0.05, 0.15), decoder=chain.DEC_TS33C,
                    max_batch=4, max_samples=n, tap_demod=True, squelch=(db, alpha, True))
    out = c.batch_process(iq)
    counts = c.fetch_squelch_counts()
    bits = c.fetch_bits().reshape(4, -1)
    assert bits.shape[1] == out.n_bits == n // D
    frames = c.fetch_frames()
    for i in range(4):
        y = odsp.fir_ccf(iq[i], taps, D)
        z = odsp.pwr_squelch(y, db, alpha, True)
        assert counts[i] == z.size and 0 < z.size < n // D
        b, _, _ = odsp.slice_bits(odsp.mag2(z), 1.0, 0.05, 0.15, 1)
        np.testing.assert_array_equal(bits[i, :z.size], b)
        assert not bits[i, z.size:].any()
        d = odec.TS33CStreamDecoder(sample_rate=20000)
        d.work([b])
        got = [f.packet() for f in frames if f.capture == i]
        assert got == [f["packet"] for f in d.frames]
    assert len(frames) >= 1
    c.close()


def test_squelch_item_exactly_on_threshold_and_no_gate():
    """alpha = 1 makes p = |y|^2: items of magnitude exactly 1 sit on a 0 dB
    threshold (passes, since mute is p < thr) and force the sequential re-run."""
    from sdr_amd import chain
    rng = np.random.default_rng(1)
    n = 5000
    x = ((rng.integers(0, 3, n) * 0.5) * np.exp(1j * np.pi / 2 * rng.integers(0, 4, n))).astype(np.complex64)
    for gate in (True, False):
        want = odsp.pwr_squelch(x, 0.0, 1.0, gate)
        c = chain.Chain(1e6, decim=1, demod=chain.DEMOD_MAG2, thr=(0.3, 0.6), tap_demod=True, max_samples=n,
                        squelch=(0.0, 1.0, gate))
        c.process_iq(x)
        assert int(c.fetch_squelch_counts()[0]) == want.size
        np.testing.assert_array_equal(c.fetch_demod(), odsp.mag2(want))
        if not gate:
            assert want.size == n
        c.close()


def test_squelch_rejects_post_filter_chain():
    from sdr_amd import _lib, chain
    with pytest.raises(_lib.SdrError):
        chain.Chain(1e6, post_taps=np.ones(3, np.float32), squelch=(-40.0, 1e-4, True))
