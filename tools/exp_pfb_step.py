"""Scratch: the channelizer bench step (k_pfb + TS33C back end + the three rule decoders) for a kernel timeline."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain
dev = torch.device("cuda", 0)
n = 1 << 26
fs = 20e6
iq = torch.randn((1, n, 2), device=dev, dtype=torch.float32) * 0.1
taps = chain.low_pass(1.0, fs, 100e3, 100e3)
c = chain.Chain(fs, taps=taps, keep_n=16, decoder=chain.DEC_TS33C, decoder_rate=fs / 64 / 16, max_samples=n, max_frames=64,
                channels=64)
rules = (chain.DEC_SOMFY, chain.DEC_BERNER, chain.DEC_MUMBI)
prof = os.environ.get("PROF", "1") == "1"
ov = os.environ.get("OVERLAP", "0") == "1"
if os.environ.get("PRE_STREAMS"):
    _keep = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ["PRE_STREAMS"]))]
c.set_overlap(ov)
def step():
    c.batch_process(iq)
    c.redecode_multi(rules, fs / 64 / 16)
for _ in range(3):
    step()
torch.cuda.synchronize()
c.profile_enable(prof)
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("step %.4f ms (profile events %s, overlap %s)" % ((time.perf_counter() - t0) / 20 * 1e3, prof, ov))
c.close()
