"""Scratch: the TFA FSK bench workload alone (front/back kernel times; SDR_FRONTEND=valu|mfma)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
v = bench.tfa_fsk_workload(torch, dev, True, steps=int(os.environ.get("STEPS", "20")))
print(os.environ.get("SDR_FRONTEND"), v["value"], "Msps front", v["roofline"]["kernel_ms"], "back", v["roofline"]["backend_kernel_ms"])
