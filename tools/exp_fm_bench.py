"""Scratch: bench.py's FM workload alone (settled clocks), one line."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
v = bench.fm_workload(torch, dev, True)
print("FM", v["ms_per_step"], "ms overlapped,", v.get("ms_per_step_one_stream"), "one stream", v["value"], "Msps")
