"""Scratch: bench.py's channelizer workload alone (one-stream pass, then overlapped) -- for a kernel-trace timeline."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
v = bench.pfb_workload(torch, dev, True)
print("pfb", v["ms_per_step"], "ms overlapped,", v.get("ms_per_step_one_stream"), "one stream", v["roofline"].get("kernel_ms_overlapped"))
