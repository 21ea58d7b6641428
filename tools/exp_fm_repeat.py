"""Scratch: the bench's FM workload (two ragged streaming chunks against the oracle) several times in one process."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bad = 0
for k in range(int(os.environ.get("REPS", "8"))):
    try:
        out = bench.fm_workload(torch, dev, False)
        print(k, "ok", out["ms_per_step"], flush=True)
    except SystemExit as e:
        bad += 1
        print(k, "FAILED:", e, flush=True)
print("failures:", bad)
