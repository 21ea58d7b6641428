"""FM-Radio chain (BASELINE.json configs[1]) against the oracle, stage by stage.
Tolerance: 1e-5 relative to the stage's peak (north_star); the float32 FIR and
discriminator stages are expected bit-identical, the de-emphasis IIR (float64
state, blocked scan on the GPU vs sequential on the CPU) within 1 ulp."""
import numpy as np
import pytest

from oracle import dsp as odsp
from sdr_amd import fm, synth

pytestmark = pytest.mark.gpu


def _cmp(got, want, exact):
    assert got.shape == want.shape
    g = got.view(np.float32) if np.iscomplexobj(got) else got
    w = want.view(np.float32) if np.iscomplexobj(want) else want
    scale = float(np.max(np.abs(w)))
    assert float(np.max(np.abs(g - w))) <= 1e-5 * scale
    if exact:
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


@pytest.mark.parametrize("fs,n", [(2.4e6, 1 << 20), (2e6, 300000)])
def test_fm_chain_stages(fs, n):
    iq = synth.fm_iq(n, fs, f_off=-200e3, seed=98)
    want = odsp.fm_radio_chain(iq, samp_rate=fs, shift=200e3, volume=0.8)
    with fm.FmRadio(samp_rate=fs, shift=200e3, volume=0.8, max_samples=n) as r:
        na = r.process(iq)
        got = [r.fetch(s) for s in range(5)]
    assert na == want[4].size > 1000
    for s in range(5):
        _cmp(got[s], want[s], exact=(s <= 2))
    # the 1 kHz test tone survives: audio is not silent
    assert float(np.std(got[4][2000:])) > 0.05


def test_fm_chain_streaming_chunks_equal_one_shot():
    """FM-Radio/FM-Radio.grc:2903-2964 is a live stream: ragged chunks (inside a decimation period of every
    stage, shorter than the filters, empty) through one object give the oracle's stage outputs, and reset()
    really returns to stream start.  FIR / discriminator stages bit-identical, IIR / resampler <= 1e-5."""
    fs, n = 2.4e6, 600000
    iq = synth.fm_iq(n, fs, f_off=-200e3, seed=7)
    want = odsp.fm_radio_chain(iq, samp_rate=fs, shift=200e3, volume=0.8)
    cuts = [0, 1, 3, 4, 5, 39, 40, 41, 1000, 1000, 77777, 77778, 300001, 512345, n]
    with fm.FmRadio(samp_rate=fs, shift=200e3, volume=0.8, max_samples=n) as r:
        for rep in range(2):
            parts = [[] for _ in range(5)]
            total = 0
            for a, b in zip(cuts[:-1], cuts[1:]):
                total += r.process(iq[a:b])
                for s in range(5):
                    parts[s].append(r.fetch(s))
            got = [np.concatenate(p) for p in parts]
            assert total == want[4].size
            for s in range(5):
                _cmp(got[s], want[s], exact=(s <= 2))
            r.reset()
        # one-shot on the same object after reset(): identical to the chunked audio
        r.process(iq)
        one = r.fetch(4)
        _cmp(one, got[4], exact=False)
        assert float(np.max(np.abs(one - got[4]))) <= 1e-6 * float(np.max(np.abs(one)))


def test_fm_audio_fir_vector_form(monkeypatch):
    """The audio FIR has two forms (matrix cores; vector ALUs when the tile does not fit LDS or the filter is
    short): force the second (SDR_FM_AUDIO=valu, read at create) and hold it to the same bits."""
    fs, n = 2e6, 300000
    iq = synth.fm_iq(n, fs, f_off=-200e3, seed=5)
    want = odsp.fm_radio_chain(iq, samp_rate=fs, shift=200e3, volume=0.8)
    monkeypatch.setenv("SDR_FM_AUDIO", "valu")
    with fm.FmRadio(samp_rate=fs, shift=200e3, volume=0.8, max_samples=n) as r:
        r.process(iq[:123457])
        a = r.fetch(2)
        r.process(iq[123457:])
        got = np.concatenate([a, r.fetch(2)])
    _cmp(got, want[2], exact=True)


def test_fm_chain_repeats_bit_for_bit():
    """The same two ragged chunks again and again, a large call in between to disturb the caches: every stage must
    repeat bit for bit.  (k_frontend_mfma2 once read its LDS copy of the rotator table before every wave had
    written its share: about one run in twenty had a few thousand wrong samples in stage 0.)"""
    torch = pytest.importorskip("torch")
    fs, m = 2.4e6, 1 << 21
    iq = synth.fm_iq(m, fs, f_off=-200e3, seed=7)
    big = torch.from_numpy(iq.view(np.float32)).to("cuda:0").repeat(8, 1)
    half = m // 2 + 4321
    with fm.FmRadio(samp_rate=fs, shift=200e3, max_samples=8 * m) as r:
        ref = None
        for it in range(40):
            r.process(big)
            r.reset()
            outs = []
            for chunk in (iq[:half], iq[half:]):
                r.process(chunk)
                outs.append([r.fetch(s).copy() for s in range(5)])
            if ref is None:
                ref = outs
                continue
            for c in range(2):
                for s in range(5):
                    assert np.array_equal(ref[c][s].view(np.uint32), outs[c][s].view(np.uint32)), (it, c, s)


def test_fm_overlap_mode_gives_the_same_audio():
    """sdr_fm_set_overlap: the audio half of call k on a second stream beside the channel filter of call k+1.  Many
    back-to-back calls without a fetch in between (device input, so nothing on the host orders them), then the
    concatenated audio must equal, bit for bit, what the same calls give on one stream -- for every call, read from
    a per-call copy made on the device behind the call."""
    torch = pytest.importorskip("torch")
    fs, m = 2.4e6, 1 << 20
    iq = synth.fm_iq(m, fs, f_off=-200e3, seed=9)
    x = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
    cuts = [0, 100001, 100001, 350000, 350007, 700000, m]
    res = {}
    for ov in (False, True):
        with fm.FmRadio(samp_rate=fs, shift=200e3, max_samples=m) as r:
            r.set_overlap(ov)
            audio = []
            for rep in range(3):                       # the stream continues across the repeats (state carried)
                for a, b in zip(cuts[:-1], cuts[1:]):
                    r.process(x[2 * a:2 * b])
                    audio.append(r.fetch(4).copy())    # fetch orders both streams; the next call overlaps again
            # and a run of calls with no fetch in between: only the last call's audio is read
            for a, b in zip(cuts[:-1], cuts[1:]):
                r.process(x[2 * a:2 * b])
            audio.append(r.fetch(4).copy())
            audio.append(r.fetch(2).copy())
            res[ov] = audio
    assert len(res[False]) == len(res[True])
    for p, q in zip(res[False], res[True]):
        assert p.shape == q.shape and np.array_equal(p.view(np.uint32), q.view(np.uint32))


@pytest.mark.parametrize("bad", [np.inf, np.nan])
def test_fm_nonfinite_sample_stays_local_on_the_matrix_core_stages(bad):
    """An Inf / NaN input sample through the chain: stage A (k_frontend_mfma2: rotation + 131 taps / 4) is
    non-finite exactly where the oracle's is and equal word for word elsewhere; the audio FIR (k_fir_fff_mfma,
    771 taps / 10) is checked against the oracle's filter run over the GPU's OWN discriminator output, which
    isolates it from how an arctangent treats NaN: the non-finite outputs are exactly those whose window holds a
    non-finite discriminator sample, the others equal bit for bit.  (Round 2 spoiled up to 15 neighbours.)"""
    fs, n = 2e6, 300000
    iq = synth.fm_iq(n, fs, f_off=-200e3, seed=11)
    iq[123456] = bad
    want = odsp.fm_radio_chain(iq, samp_rate=fs, shift=200e3, volume=0.8)
    with fm.FmRadio(samp_rate=fs, shift=200e3, volume=0.8, max_samples=n) as r:
        r.process(iq)
        y0, q, a2 = r.fetch(0), r.fetch(1), r.fetch(2)
    w0 = want[0].view(np.float32)
    g0 = y0.view(np.float32)
    nf = ~np.isfinite(w0)
    assert 30 <= nf.sum() <= 2 * 34
    assert np.array_equal(~np.isfinite(g0), nf)
    assert np.array_equal(g0[~nf].view(np.uint32), w0[~nf].view(np.uint32))
    audio_rate = 500e3 / 10
    taps = odsp.low_pass(1.0, 500e3, audio_rate / 2 - audio_rate / 32, audio_rate / 32, odsp.WIN_HAMMING)
    ref = odsp.fir_fff(q, taps, 10)
    nf2 = ~np.isfinite(ref)
    if np.isnan(bad) and np.isfinite(q).all():
        nf2[:] = False    # the arctangent's comparisons turned the NaNs into finite angles: nothing reaches the audio FIR
    else:
        assert (~np.isfinite(q)).sum() >= 1 and 70 <= nf2.sum() <= 90
    assert np.array_equal(~np.isfinite(a2), nf2)
    assert np.array_equal(a2[~nf2].view(np.uint32), ref[~nf2].view(np.uint32))


def test_fm_fused_tail_equals_the_two_kernels(monkeypatch):
    """De-emphasis + audio resampler in one launch (k_deemph_resample, the default) against the two separate
    kernels (SDR_FM_TAIL=split, read at create): de-emphasised floats and audio identical bit for bit, over ragged
    chunks that end inside a workgroup's halo, exactly on a workgroup boundary (4048) and one float beyond it."""
    fs = 2e6
    # audio-rate floats per chunk = chunk / 40: 4048 * 40 = 161920 input samples per workgroup span
    cuts = [0, 40 * 4048, 40 * 4048 + 40, 40 * 4048 + 1960, 40 * 12144 + 17, 40 * 12144 + 57, 700001, 800000]
    iq = synth.fm_iq(cuts[-1], fs, f_off=-200e3, seed=21)
    res = {}
    for mode in ("fused", "split"):
        if mode == "split":
            monkeypatch.setenv("SDR_FM_TAIL", "split")
        else:
            monkeypatch.delenv("SDR_FM_TAIL", raising=False)
        with fm.FmRadio(samp_rate=fs, shift=200e3, volume=0.8, max_samples=cuts[-1]) as r:
            a3, a4 = [], []
            for a, b in zip(cuts[:-1], cuts[1:]):
                r.process(iq[a:b])
                a3.append(r.fetch(3))
                a4.append(r.fetch(4))
            res[mode] = (np.concatenate(a3), np.concatenate(a4))
    for k in range(2):
        assert res["fused"][k].size == res["split"][k].size > 10000
        assert np.array_equal(res["fused"][k].view(np.uint32), res["split"][k].view(np.uint32))
    want = odsp.fm_radio_chain(iq, samp_rate=fs, shift=200e3, volume=0.8)
    _cmp(res["fused"][0], want[3], exact=False)
    _cmp(res["fused"][1], want[4], exact=False)
