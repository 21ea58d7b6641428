"""Scratch: the FM chain over the same two ragged chunks again and again; every stage's output must repeat bit for bit."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import fm, synth
dev = torch.device("cuda", 0)
fs, m = 2.4e6, 1 << 22
n = int(os.environ.get("MAXN", str(1 << 26)))
iq = synth.fm_iq(m, fs, f_off=200e3, seed=98)
r = fm.FmRadio(samp_rate=fs, shift=200e3, max_samples=n)
st = torch.cuda.Stream(device=dev)
r.set_stream(st.cuda_stream)
half = m // 2 + 12345
big = None
if os.environ.get("BIG", "1") == "1":
    x1 = torch.from_numpy(iq.view(np.float32)).to(dev)
    big = x1.repeat(n // m, 1) if x1.dim() == 2 else x1.repeat(n // m)
ref = None
bad = 0
for it in range(int(os.environ.get("REPS", "30"))):
    if big is not None:
        for _ in range(2):
            r.process(big)          # what the bench does before the check
    r.reset()
    outs = []
    for chunk in (iq[:half], iq[half:]):
        if os.environ.get("DEVIN") == "1":
            chunk = torch.from_numpy(chunk.view(np.float32).reshape(-1, 2)).to(dev)
            torch.cuda.synchronize()
        r.process(chunk)
        outs.append([r.fetch(s).copy() for s in range(5)])
    if ref is None:
        ref = outs
        continue
    for c in range(2):
        for s in range(5):
            a, b = ref[c][s], outs[c][s]
            if a.shape != b.shape or not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                d = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32)) if a.shape == b.shape else np.array([-1])
                print("iter %d chunk %d stage %d: %d of %d words differ, first %d last %d" % (it, c, s, d.size, a.size, d[0], d[-1]), flush=True)
                cuts = np.flatnonzero(np.diff(d) > 64)
                starts = np.concatenate([[d[0]], d[cuts + 1]])
                ends = np.concatenate([d[cuts], [d[-1]]])
                print("   %d runs (gaps > 64 words); first runs (start, end, words): %s" % (starts.size, [(int(a_), int(b_), int(b_ - a_ + 1)) for a_, b_ in zip(starts[:12], ends[:12])]), flush=True)
                bad += 1
                break
        else:
            continue
        break
print("iterations with a difference:", bad)
