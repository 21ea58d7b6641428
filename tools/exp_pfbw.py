"""Scratch: the wideband bench workload alone."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
v = bench.pfb_workload(torch, dev, False)
print(v["value"], "Msps", v["ms_per_step"], "ms/step; k_pfb", v["roofline"]["kernel_ms"], "back (4 decoders)", v["roofline"]["backend_kernel_ms"], v["verified"])
