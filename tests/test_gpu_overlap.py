"""sdr_set_overlap (include/sdrhip.h): the back end of batch k on the result stream beside the front end of batch
k+1.  Bar: sliced bytes, frame records, counts and every later call on the handle bit-identical to the same
handle type without overlap."""
import ctypes as C

import numpy as np
import pytest

from sdr_amd import _lib, chain, synth
from tests.test_zz_no_alloc import _batch

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FS, D, MAXF = 1e6, 50, 4


def _chain(max_batch, n, **kw):
    taps = chain.low_pass(1.0, FS, 20e3, 40e3)
    args = dict(decim=D, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_batch=max_batch, max_frames=MAXF,
                max_samples=n)
    args.update(kw)
    return chain.Chain(FS, **args)


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(x.shape[0], x.shape[1], 2)).to("cuda:0")


def _exports(c, x, outs):
    """one batch and its device-side exports into fresh tensors appended to outs; no host synchronisation"""
    ncap = x.shape[0]
    c.batch_process(x)
    nb = int(c.out.n_bits)
    bits = torch.full((ncap * nb,), 0xAB, dtype=torch.uint8, device=x.device)
    rec = torch.full((ncap * MAXF * C.sizeof(_lib.Frame),), 0xCD, dtype=torch.uint8, device=x.device)
    cnt = torch.full((ncap,), -7, dtype=torch.int32, device=x.device)
    assert c.export_bits(bits) == ncap * nb
    c.export_frames(rec, cnt)
    outs.append((bits, rec, cnt))


def test_overlapped_batches_match_serial_ones():
    n = 1 << 18
    shapes = [(24, n, 11), (24, n, 12), (9, 200000, 13), (24, n, 14), (1, n, 15), (24, n, 16), (24, n, 17)]
    xs = [_dev(_batch(ncap, m, FS, seed)) for ncap, m, seed in shapes]
    torch.cuda.synchronize()
    with _chain(24, n) as ref, _chain(24, n) as c:
        want, got = [], []
        for x in xs:
            _exports(ref, x, want)
        ref.synchronize()
        assert c.result_stream() != 0
        own = c.result_stream()
        c.set_overlap(True)
        assert c.result_stream() not in (0, own)
        for x in xs:                      # queued back to back: batch k's back end runs beside batch k+1's front end
            _exports(c, x, got)
        c.synchronize()                   # joins the result stream
        total = 0
        for k, ((wb, wr, wc), (gb, gr, gc)) in enumerate(zip(want, got)):
            assert torch.equal(wc, gc), k
            assert torch.equal(wb, gb), k
            assert torch.equal(wr, gr), k
            total += int(gc.sum())
        assert total >= sum(s[0] for s in shapes)     # the captures do carry frames
        # host fetches join by themselves: the last batch's frames and bytes
        fr = c.fetch_frames()
        assert len(fr) == int(got[-1][2].sum())
        assert [f.packet() for f in fr] == [f.packet() for f in ref.fetch_frames()]
        assert np.array_equal(c.fetch_bits(), ref.fetch_bits())
        # a batch, a host fetch, a batch: each step ends the overlap and starts it again
        for x in xs[:3]:
            c.batch_process(x)
            ref.batch_process(x)
            assert np.array_equal(c.fetch_bits(), ref.fetch_bits())
            assert [(f.capture, f.packet()) for f in c.fetch_frames()] == [(f.capture, f.packet()) for f in ref.fetch_frames()]
        # redecode on the result stream, then the allocation-free entry point and a streaming call on the same handle
        c.batch_process(xs[0])
        c.redecode(chain.DEC_TS33C, 20000.0)
        ref.batch_process(xs[0])
        assert [(f.capture, f.packet()) for f in c.fetch_frames()] == [(f.capture, f.packet()) for f in ref.fetch_frames()]
        c.batch_process(xs[1])
        outs = []
        for h in (c, ref):
            ncap, nb = xs[3].shape[0], int(h.out.n_bits)
            bits = torch.zeros(ncap * nb, dtype=torch.uint8, device="cuda:0")
            rec = torch.zeros(ncap * MAXF * C.sizeof(_lib.Frame), dtype=torch.uint8, device="cuda:0")
            cnt = torch.zeros(ncap, dtype=torch.int32, device="cuda:0")
            h.batch_process_static(xs[3], bits, rec, cnt)
            h.synchronize()
            outs.append((bits, rec, cnt))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        c.set_overlap(False)
        assert c.result_stream() == own


def test_overlap_then_streaming_calls():
    """streaming calls on a handle whose last batches were overlapped: they join first, same frames as without"""
    n = 1 << 18
    x = _batch(2, n, FS, 21)
    seen = []
    with _chain(1, n) as ref, _chain(1, n) as c:
        c.set_overlap(True)
        for h in (c, ref):
            for _ in range(3):
                h.batch_process(_dev(x[:1]))
            frames = []
            for p in np.array_split(x[1], 7):
                h.process_iq(p)
                frames += [(f.status, f.packet()) for f in h.fetch_frames()]
            seen.append(frames)
    assert seen[0] == seen[1] and len(seen[1]) >= 1


def test_overlap_on_long_rows_and_the_channelizer():
    """rows long enough for the segment- and piece-parallel back end (three launches on the result stream), and
    a channelizer handle with the multi-decoder launch behind each batch"""
    fs, n = 2e6, 1 << 20
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    rng = np.random.default_rng(3)
    xs = []
    for k in range(3):
        sym = []
        for _ in range(6):
            sym += synth.ts33c_symbols(synth.ts33c_bits(synth.ts33c_packet(**synth.random_ts33c_fields(rng))), rng)
        env = synth.envelope(sym, fs, lead_in_us=5000.0, total=n)
        xs.append(_dev(np.stack([synth.ook_iq(env, fs, f_off=0.0, amp=0.7, sigma=0.02, rng=rng) for _ in range(2)])))
    kw = dict(decim=1, taps=taps, keep_n=100, decoder=chain.DEC_TS33C, decoder_rate=fs / 100, max_batch=2, max_frames=16,
              max_samples=n)
    with chain.Chain(fs, **kw) as ref, chain.Chain(fs, **kw) as c:
        c.set_overlap(True)
        want, got = [], []
        for h, outs in ((ref, want), (c, got)):
            for x in xs:
                h.batch_process(x)
                rec = torch.zeros(2 * 16 * C.sizeof(_lib.Frame), dtype=torch.uint8, device="cuda:0")
                cnt = torch.zeros(2, dtype=torch.int32, device="cuda:0")
                h.export_frames(rec, cnt)
                outs.append((rec, cnt))
            h.synchronize()
        for (wr, wc), (gr, gc) in zip(want, got):
            assert torch.equal(wc, gc) and torch.equal(wr, gr)
            assert int(wc.sum()) >= 2
        assert np.array_equal(c.fetch_bits(), ref.fetch_bits())

    FSW, M, keep = 20e6, 64, 16
    nw = 1 << 22
    rate = FSW / M / keep
    proto = chain.low_pass(1.0, FSW, 100e3, 100e3)
    pkt = synth.ts33c_packet(channel=1, rolling_code=9, temp_tenths=215, humidity=40, filler=(1, 2, 3, 4, 5))
    caps = []
    for seed, chn in ((1, 5), (2, 9), (3, 20)):
        r2 = np.random.default_rng(seed)
        em = [(chn * FSW / M, synth.ts33c_symbols(synth.ts33c_bits(pkt), r2) * 2, 5000.0),
              ((chn + 7) * FSW / M, synth.mumbi_symbols(bytes([0x12, 0x34, 0x56, 0x78, 0x80]), repeats=4), 6000.0)]
        caps.append(torch.from_numpy(synth.wideband_capture(nw, FSW, em, seed=seed).view(np.float32).reshape(1, nw, 2)).to("cuda:0"))
    rules = (chain.DEC_SOMFY, chain.DEC_BERNER, chain.DEC_MUMBI)
    kwp = dict(taps=proto, keep_n=keep, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=nw, max_frames=16, channels=M)
    with chain.Chain(FSW, **kwp) as ref, chain.Chain(FSW, **kwp) as c:
        c.set_overlap(True)
        for h in (ref, c):
            h._seen = []
        for rounds in range(2):
            for x in caps:                # overlapped: nothing joins between these calls
                c.batch_process(x)
                c.redecode_multi(rules, rate)
        for x in caps:
            for h in (ref, c):
                h.batch_process(x)
                h.redecode_multi(rules, rate)
                h._seen.append([(f.capture, f.status, f.packet()) for f in h.fetch_frames()])
                for k in range(len(rules)):
                    h._seen.append([(f.capture, f.status, f.packet()) for f in h.fetch_frames_multi(k)])
        assert c._seen == ref._seen
        assert sum(len(s) for s in ref._seen) >= 6


def test_profile_every_nth_call_and_redecode_inherits():
    """sdr_profile_enable(h, N): every N-th processing call is timed, the sdr_redecode behind a timed call with it;
    results do not depend on the setting."""
    n, ncap = 1 << 17, 6
    iq = _batch(ncap, n, FS, seed=5)
    x = _dev(iq)
    with _chain(ncap, n) as c:
        c.batch_process(x)
        want = [(f.capture, f.status, f.packet()) for f in c.fetch_frames()]
        c.profile_enable(True, 3)
        for k in range(9):
            c.batch_process(x)
            c.redecode(chain.DEC_TS33C)
        front, back = c.profile_read()
        # calls 0, 3, 6 are timed: a (front, back) slot each, and one slot for the redecode behind each of them
        assert len(front) == 6 and len(back) == 6
        assert all(v > 0 for v in front[0::2]) and all(v >= 0 for v in back)
        assert [(f.capture, f.status, f.packet()) for f in c.fetch_frames()] == want
        c.profile_enable(True)
        for k in range(4):
            c.batch_process(x)
        front, back = c.profile_read()
        assert len(front) == 4
        c.profile_enable(False)
        c.batch_process(x)
        assert len(c.profile_read()[0]) == 0
