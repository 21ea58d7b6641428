"""Scratch: front-end kernel time of one D=1 shape under SDR_FRONTEND=win|mfma|valu.
usage: exp_win.py ntaps demod(mag2|quad) keep_n ncap log2n"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain
if len(sys.argv) < 6:
    sys.argv = [sys.argv[0]] + os.environ.get("WIN_ARGS", "27 quad 1 64 20 win").split()
nt, demod, keep, ncap, lg = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = 1 << lg
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
iq = 0.3 * torch.randn((ncap, n, 2), generator=g, device=dev, dtype=torch.float32)
cu8 = bool(os.environ.get("WIN_CU8"))
if cu8:
    iq = torch.clamp(torch.round(iq * 128.0 + 127.4), 0, 255).to(torch.uint8)
rng = np.random.default_rng(nt)
taps = (rng.standard_normal(nt) / nt ** 0.5).astype(np.float32)
for mode in sys.argv[6:] or ["win", "mfma", "valu"]:
    os.environ["SDR_FRONTEND"] = mode
    c = chain.Chain(2e6, decim=1, taps=taps, demod=chain.DEMOD_QUAD if demod == "quad" else chain.DEMOD_MAG2,
                    thr=(-0.2, 0.2) if demod == "quad" else (0.05, 0.15), keep_n=keep, max_batch=ncap, max_frames=4, max_samples=n,
                    center_freq=float(os.environ.get("WIN_FC", "0")), input_format="cu8" if cu8 else "fc32",
                    tap_demod=bool(os.environ.get("WIN_TAP")))
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        c.batch_process(iq)
    torch.cuda.synchronize()
    c.profile_enable(True)
    for _ in range(20):
        c.batch_process(iq)
    torch.cuda.synchronize()
    f, b = c.profile_read()
    ms = float(np.mean(f))
    print("%-5s ntaps %3d %s keep %3d: front %.4f ms  %.1f Gsps  %.1f TF  %.0f GB/s" % (mode, nt, demod, keep, ms, ncap * n / ms / 1e6, 4.0 * nt * ncap * n / ms / 1e9, 8.0 * ncap * n / ms / 1e6), flush=True)
    c.close()
