"""Scratch: where a graph replay of sdr_batch_process_static differs from sdr_batch_process + exports."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain, _lib
from tests.test_zz_no_alloc import _batch, _reference
fs, D, max_frames = 1e6, 50, 4
taps = chain.low_pass(1.0, fs, 20e3, 40e3)
dev = torch.device("cuda", 0)
ncap, n = 24, 1 << 18
kw = dict(decim=D, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_batch=32, max_frames=max_frames, max_samples=n)
ref, c = chain.Chain(fs, **kw), chain.Chain(fs, **kw)
batches = [torch.from_numpy(_batch(ncap, n, fs, seed).view(np.float32).reshape(ncap, n, 2)).to(dev) for seed in (31, 32)]
x = batches[0].clone()
nb = n // D
bits = torch.zeros(ncap * nb, dtype=torch.uint8, device=dev)
rec = torch.zeros(ncap * max_frames * C.sizeof(_lib.Frame), dtype=torch.uint8, device=dev)
cnt = torch.zeros(ncap, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
c.set_stream(st.cuda_stream)
torch.cuda.synchronize()
mode = os.environ.get("MODE", "graph")
if mode == "graph":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        c.batch_process_static(x, bits, rec, cnt)
    torch.cuda.synchronize()
for k, b in enumerate(batches + batches[:1]):
    x.copy_(b)
    torch.cuda.synchronize()
    if mode == "graph":
        g.replay()
    else:
        with torch.cuda.stream(st):
            c.batch_process_static(x, bits, rec, cnt)
    torch.cuda.synchronize()
    want_nb, want_bits, want_rec, want_cnt = _reference(ref, b, max_frames)
    got = rec.cpu().numpy()
    d = np.flatnonzero(got != want_rec)
    print("call", k, "counts equal", np.array_equal(cnt.cpu().numpy(), want_cnt), "bits equal", np.array_equal(bits.cpu().numpy(), want_bits),
          "record bytes differing:", d.size, "offsets in record:", sorted(set((d % 96).tolist()))[:20], "records:", sorted(set((d // 96).tolist()))[:10])
    if d.size:
        i = d[0] // 96
        print("   got ", got[96 * i:96 * i + 96].tolist())
        print("   want", want_rec[96 * i:96 * i + 96].tolist())
