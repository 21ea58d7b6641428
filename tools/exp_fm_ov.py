"""Scratch: the bench's FM workload, one stream and overlapped."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
for _ in range(2):
    v = bench.fm_workload(torch, dev, True)
    print("FM one stream %.4f ms, overlapped %.4f ms, frac %.3f" % (v["ms_per_step_one_stream"], v["ms_per_step"], v["roofline"]["frac"]), flush=True)
