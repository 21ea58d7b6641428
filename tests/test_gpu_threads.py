"""SURVEY 8b, threading: one handle = one stream, calls on a handle serialised by its caller, different handles
usable from different threads.  Four threads, each with its own chain of a different shape, run batches,
streaming calls and fetches at the same time; every thread must see what the same work gives when run alone."""
import threading

import numpy as np
import pytest

from sdr_amd import chain
from tests.test_zz_no_alloc import _batch

pytestmark = pytest.mark.gpu


def _work(kind, x, rounds):
    """-> list of (sliced bytes, frames) per round for one chain shape"""
    fs = 1e6
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    n = x.shape[1]
    kw = {"d50": dict(decim=50, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C),
          "d10": dict(decim=10, center_freq=100e3, taps=taps, keep_n=5, decoder=chain.DEC_TS33C, decoder_rate=2e4),
          "d1": dict(decim=1, center_freq=100e3, taps=taps, keep_n=50, decoder=chain.DEC_TS33C, decoder_rate=2e4),
          "quad": dict(decim=4, taps=taps, demod=chain.DEMOD_QUAD, demod_gain=2.0, thr=(-0.3, 0.3), decoder=chain.DEC_BERNER)}[kind]
    out = []
    with chain.Chain(fs, max_batch=x.shape[0], max_frames=8, max_samples=n, **kw) as c:
        for r in range(rounds):
            c.batch_process(x)
            out.append((c.fetch_bits().copy(), [(f.capture, f.status, f.packet()) for f in c.fetch_frames()]))
    # the same input as one stream in ragged chunks on a fresh handle
    with chain.Chain(fs, max_batch=1, max_frames=8, max_samples=n, **kw) as s:
        frames = []
        for p in np.array_split(x[0], 5):
            s.process_iq(p)
            frames += [(f.status, f.packet()) for f in s.fetch_frames()]
        out.append((s.fetch_bits().copy(), frames))
    return out


def test_four_handles_on_four_threads():
    kinds = ["d50", "d10", "d1", "quad"]
    xs = {k: _batch(6, 1 << 18, 1e6, 40 + i) for i, k in enumerate(kinds)}
    alone = {k: _work(k, xs[k], 3) for k in kinds}
    got, errs = {}, []

    def run(k):
        try:
            got[k] = _work(k, xs[k], 3)
        except Exception as e:       # noqa: BLE001 -- reported below, in the main thread
            errs.append((k, repr(e)))

    for _ in range(3):
        got.clear()
        th = [threading.Thread(target=run, args=(k,)) for k in kinds]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for k in kinds:
            assert len(got[k]) == len(alone[k])
            for (b0, f0), (b1, f1) in zip(alone[k], got[k]):
                assert np.array_equal(b0, b1), k
                assert f0 == f1, k
    assert any(f for _, f in alone["d50"])     # frames were decoded at all
