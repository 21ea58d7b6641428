"""Scratch: the mixed somfy/berner/mumbi bench workload alone (front/back kernel times)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
v = bench.mixed_emitters_workload(torch, dev, True)
print(v["value"], "Msps front", v["roofline"]["kernel_ms"], "back (3 decoders)", v["roofline"]["backend_kernel_ms"])
