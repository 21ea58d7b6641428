"""Scratch: the bench's secondary workloads with sdr_set_overlap off / on, alternating in one process."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
keep = []
if os.environ.get("KEEP_MAIN"):
    from sdr_amd import chain
    taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
    for i in range(int(os.environ["KEEP_MAIN"])):
        c = chain.Chain(bench.FS, decim=bench.DECIM, center_freq=bench.F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_batch=64,
                        max_frames=4, max_samples=bench.N_PER_CAPTURE)
        c.set_overlap(True)
        keep.append(c)
for rep in range(2):
    for ov in (False, True):
        bench.OVERLAP = ov
        res = {"r": bench.r_shape_workload(torch, dev, True), "pfb": bench.pfb_workload(torch, dev, True),
               "tfa": bench.tfa_fsk_workload(torch, dev, True), "mixed": bench.mixed_emitters_workload(torch, dev, True)}
        print("overlap %-5s " % ov + "  ".join("%s %.4f (front %.4f back %.4f)" % (k, v["ms_per_step"], v["roofline"]["kernel_ms"],
                                                                                     v["roofline"]["backend_kernel_ms"]) for k, v in res.items()), flush=True)
