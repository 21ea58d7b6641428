#!/usr/bin/env python3
"""bench.py -- receive-chain throughput on MI355X (one process per GPU).

Workload (config.workload = "ook_batch_ts33c_xlating"): BASELINE.json configs[3]
per-GPU share (a batch of synthetic OOK/PWM captures, 2^18 complex64 samples
each at 1 Msps) through the chain of configs[0]: freq-xlating FIR (61 taps,
fc 20 kHz / tw 40 kHz Hamming designed at fs, centre +100 kHz, decimation 50)
-> complex_to_mag_squared -> threshold .05/.15 -> slicer -> TS33C stream
decoder, all on the GPU.  A step = one sdr_batch_process call over the whole
batch (front-end kernel + back-end kernel), inputs resident in HBM.

Metric: input IQ Msps, whole job (all ranks).  Weak scaling: every rank owns
`--captures` captures; with N>1 the decoded frame records and counts are
all-gathered over RCCL inside the timed step (the only collective).

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 1e6
N_PER_CAPTURE = 1 << 18
DECIM = 50
F_OFF = 100e3
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# sdr_set_overlap: the back end of step k (latency bound: one workgroup per capture or row) runs on the handle's
# second stream beside the front end of step k+1.  It pays where the back end's workgroups are few (the 64-row
# workloads: -4 % R shape, -9 % TFA, -22 % channelizer step) and not where 1250 of them fill the chip's LDS (the
# headline and cu8 batches: same step, front-end kernel 7 % slower), so those two stay on one stream.  A secondary
# workload is timed twice, one stream first (its kernel times feed `roofline`), then overlapped (`ms_per_step`,
# `value`); every step of either pass does all of its work and each pass ends with a device-wide synchronize.
# SDR_BENCH_OVERLAP=0 skips the second pass.
OVERLAP = os.environ.get("SDR_BENCH_OVERLAP", "1") != "0"
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector peak = f32-input MFMA peak
BYTES_PER_SAMPLE = 8.0 + 1.0 / DECIM  # complex64 in + sliced byte out


def make_batch(torch, dev, ncap, seed, FS=FS, N_PER_CAPTURE=N_PER_CAPTURE, F_OFF=F_OFF):
    """Synthetic captures built on the GPU: host makes the 0/1 envelopes of two
    TS33C frames per capture (random payloads), the GPU applies carrier + AWGN."""
    from sdr_amd import synth
    rng = np.random.default_rng(seed)
    env = np.zeros((ncap, N_PER_CAPTURE), np.uint8)
    pkts = []
    for i in range(ncap):
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        pkts.append(pkt)
        sym = []
        for _ in range(3):
            sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
        env[i] = synth.envelope(sym, FS, lead_in_us=5000.0 + 200.0 * rng.random(), total=N_PER_CAPTURE)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    iq = torch.empty((ncap, N_PER_CAPTURE, 2), dtype=torch.float32, device=dev)
    t = torch.arange(N_PER_CAPTURE, dtype=torch.float64, device=dev)
    ph0 = torch.from_numpy(rng.random(ncap) * 2 * np.pi).to(dev)
    step = 64
    for a in range(0, ncap, step):
        b = min(ncap, a + step)
        e = torch.from_numpy(env[a:b]).to(dev).to(torch.float32) * 0.7
        ph = (2 * np.pi * F_OFF / FS) * t[None, :] + ph0[a:b, None]
        iq[a:b, :, 0] = e * torch.cos(ph).to(torch.float32)
        iq[a:b, :, 1] = e * torch.sin(ph).to(torch.float32)
        iq[a:b] += 0.02 * torch.randn((b - a, N_PER_CAPTURE, 2), generator=g, device=dev, dtype=torch.float32)
    return iq, pkts


def timed_repeat(fn, min_s=1.0):
    """Run fn() until at least min_s seconds of it have been timed -> (first result, repetitions, seconds)."""
    first, reps, used = None, 0, 0.0
    while used < min_s:
        t0 = time.perf_counter()
        r = fn()
        used += time.perf_counter() - t0
        if reps == 0:
            first = r
        reps += 1
    return first, reps, used


def host_cores():
    """CPU threads this job may use: the affinity mask, capped by the box's per-GPU share (OMP_NUM_THREADS is
    set to it there; os.cpu_count() shows the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_worker(path, first, count, budget_s):
    """One host core of the all-cores CPU baseline (started by cpu_baseline as `bench.py --cpu-worker ...`, a
    fresh process that never touches the GPU): the oracle chain + numpy TS33C decoder over its share of the
    sampled captures, round and round until the budget is used.  Prints {"done", "seconds", "frames"}."""
    from oracle import decoders as odec
    from oracle import dsp as odsp
    caps = np.load(path, mmap_mode="r")
    taps = np.load(path + ".taps.npy")
    done, used, frames, i = 0, 0.0, 0, 0
    while used < budget_s:
        x = np.ascontiguousarray(caps[first + i % count]).view(np.complex64).reshape(-1)
        t0 = time.perf_counter()
        bits = odsp.ook_chain(x, taps, DECIM, 1.0, 0.05, 0.15, 1, fc=F_OFF, fs=FS)
        d = odec.TS33CStreamDecoder(sample_rate=20000)
        d.work([bits])
        used += time.perf_counter() - t0
        frames += len(d.frames)
        done += 1
        i += 1
    print(json.dumps({"done": done, "seconds": used, "frames": frames}))


def worker_env():
    """Environment of the CPU-baseline worker processes: they never touch the GPU, so a profiler wrapped around
    this command (rocprofv3 sets LD_PRELOAD / ROCP* / ROCPROF* for its tool library) is not inherited -- its
    library would otherwise run in every worker and write trace files of its own into the same directory."""
    env = {k: v for k, v in os.environ.items() if not (k.startswith("ROCP") or k.startswith("ROCPROF") or k == "HSA_TOOLS_LIB")}
    if "LD_PRELOAD" in env:
        keep = [x for x in env["LD_PRELOAD"].replace(":", " ").split() if "rocprof" not in os.path.basename(x).lower()]
        if keep:
            env["LD_PRELOAD"] = ":".join(keep)
        else:
            del env["LD_PRELOAD"]
    env["OMP_NUM_THREADS"] = "1"
    return env


def cpu_sample(iq_gpu):
    """the captures the CPU baseline runs over: two per host core from the head of this rank's batch, on the host"""
    return iq_gpu[:min(iq_gpu.shape[0], 2 * host_cores())].cpu().numpy()


def cpu_baseline(sample, taps, budget_s=None):
    """The oracle (C restatement of the GNU Radio blocks + numpy restatement of TS33C/stream_decoder.py) over a
    bounded sample of the same captures: one host core (the scalar port, `value`) and ALL the job's host
    cores (one worker process per core over disjoint captures, `all_cores`).  A stand-in for the reference
    flowgraph (GNU Radio is not installable here), reported beside the GPU number."""
    import subprocess
    import tempfile
    from oracle import decoders as odec
    from oracle import dsp as odsp
    if budget_s is None:
        budget_s = float(os.environ.get("SDR_BENCH_CPU_BUDGET", "10"))
    cores = host_cores()
    nsample = sample.shape[0]

    def one_pass(k=[0]):
        x = sample[k[0] % nsample].view(np.complex64).reshape(-1)
        k[0] += 1
        bits = odsp.ook_chain(x, taps, DECIM, 1.0, 0.05, 0.15, 1, fc=F_OFF, fs=FS)
        d = odec.TS33CStreamDecoder(sample_rate=20000)
        d.work([bits])
        return len(d.frames)

    _, done, t_used = timed_repeat(one_pass, budget_s)
    msps = done * N_PER_CAPTURE / t_used / 1e6
    out = {"value": round(msps, 2), "unit": "Msps", "cores": 1, "kind": "port",
           "sample": "%d capture passes x 2^18 samples drawn from the same batch, oracle C chain + numpy TS33C decoder, %.1f s"
                     % (done, t_used)}
    # all cores: one fresh worker process per core (no GPU in them), disjoint captures, same time budget
    tmp = tempfile.mkdtemp(prefix="sdr_cpu_")
    path = os.path.join(tmp, "caps.npy")
    try:
        np.save(path, sample)
        np.save(path + ".taps.npy", np.asarray(taps, np.float32))
        per = max(1, nsample // cores)
        env = worker_env()
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str((w * per) % nsample),
                                   str(per), str(budget_s)], stdout=subprocess.PIPE, env=env) for w in range(cores)]
        res = []
        for p in procs:
            o, _ = p.communicate()
            if p.returncode != 0:
                raise SystemExit("cpu_baseline: a worker failed (exit %d)" % p.returncode)
            res.append(json.loads(o.decode().strip().splitlines()[-1]))
        wall = time.perf_counter() - t0
        total = sum(r["done"] for r in res)
        rate = sum(r["done"] * N_PER_CAPTURE / r["seconds"] for r in res) / 1e6
        out["all_cores"] = {"value": round(rate, 2), "unit": "Msps", "cores": cores, "kind": "port",
                            "os_cpu_count": os.cpu_count(),
                            "sample": "%d worker processes (one per core of this job's CPU share), %d capture passes in all, "
                                      "%.1f s of oracle time each, %.1f s wall including start-up" % (cores, total, budget_s, wall)}
    finally:
        for f in (path, path + ".taps.npy"):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)
    return out


def fm_workload(torch, dev, skip_cpu, n=1 << 26, fs=2.4e6, steps=10):
    """Secondary line (BASELINE.json configs[1]): FM-Radio WBFM chain at 2.4 Msps -> 48/50-resampled audio.
    2^26 samples (512 MB of IQ, twice the 256 MB Infinity Cache, so every step streams from HBM) fed as one
    chunk per step into a STREAMING chain object (block state carried from step to step, as the live
    flowgraph runs); the input is 16 repeats of a 2^22-sample synthetic capture.  Timed with events on the
    stream the kernels run on."""
    from sdr_amd import fm, synth
    m = 1 << 22
    iq = synth.fm_iq(m, fs, f_off=-200e3, seed=98)
    x1 = torch.from_numpy(iq.view(np.float32)).to(dev)
    x = x1.repeat(n // m, 1) if x1.dim() == 2 else x1.repeat(n // m)
    r = fm.FmRadio(samp_rate=fs, shift=200e3, max_samples=n)
    st = torch.cuda.Stream(device=dev)            # a real stream handle (the default stream's is 0 = "the chain's own")
    r.set_stream(st.cuda_stream)
    torch.cuda.synchronize()
    prewarm(torch, lambda: r.process(x))
    for _ in range(2):
        na = r.process(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(st)
    for k in range(steps):
        r.process(x)
        ev[k + 1].record(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gpu_ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]))
    bps = 8.0 + 4.0 * 48e3 / fs
    ach = bps * n / (gpu_ms * 1e-3) / 1e9
    # algorithmic flops per input sample of the four FIRs (real taps on complex data = 4 flop per tap, on floats 2):
    # 131-tap resampler /4, 33-tap LPF at fs/4, 771-tap audio filter /10 at fs/4, 821-tap 24/25 polyphase resampler
    fps = 4.0 * 131 / 4 + 4.0 * 33 / 4 + 2.0 * 771 / 40 + 2.0 * (821 / 24.0) * (24.0 / 25.0) / 40
    tf = fps * n / (gpu_ms * 1e-3) / 1e12
    fm_traffic, fm_traffic_src = pmc_traffic("fm chain", "samples=2^%d" % int(np.log2(n)))   # sum over the chain's kernels
    out = {"value": round(n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4), "samples": n,
           "sample_rate": fs, "audio_samples": int(na),
           "chain": "xlating->resample 1/4->lpf->quadrature_demod->audio fir /10->deemph 75us->resample 24/25 (streaming state carried)",
           "roofline": {"bound": "mfma", "kernel": "whole chain: k_frontend_mfma2 (rotate + 131 taps /4) + k_frontend_win (33 taps, "
                        "discriminator) + k_fir_fff_mfma (771 taps /10) + k_deemph_resample (de-emphasis + 24/25 resampler)",
                        "achieved": round(tf, 1), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / FP32_PEAK_TFLOPS, 4), "traffic": fm_traffic, "traffic_source": fm_traffic_src, "kernel_ms": round(gpu_ms, 4),
                        "flops_per_launch": fps * n, "hbm_achieved_gbs": round(ach, 1), "hbm_frac": round(ach / HBM_PEAK_GBS, 4),
                        "bytes_per_launch": bps * n,
                        "note": "%.0f algorithmic flop per 8.08 algorithmic bytes (SURVEY 8d) is above the fp32 ridge (157.3 TF / "
                                "8 TB/s = 19.7 flop/B): the chain is priced against the f32 FMA peak; hbm_frac is the same time "
                                "against the HBM roofline.  The stage outputs between the five kernels (2 + 1 B per input "
                                "sample, written and re-read) are not counted" % fps}}
    if OVERLAP:
        # second pass: the audio half of chunk k on a second stream beside the channel filter of chunk k+1
        # (sdr_fm_set_overlap); wall clock over the same steps, device-wide synchronize at the end
        progress("  second pass, overlapped")
        r.set_overlap(True)
        for _ in range(3):
            r.process(x)
        r.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            r.process(x)
        r.synchronize()
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / steps
        r.set_overlap(False)
        if dt2 >= dt:
            # every kernel of this chain is arithmetic-bound: on settled clocks the second stream buys launch gaps only,
            # and on some boxes not even those -- the line then stays the one-stream measurement
            out["ms_per_step_overlapped"] = round(dt2 * 1e3, 4)
            out["streams"] = "one stream (sdr_fm_set_overlap measured in a second pass: not faster here)"
    if OVERLAP and "ms_per_step_overlapped" not in out:
        out["ms_per_step_one_stream"] = out["ms_per_step"]
        out["ms_per_step"] = round(dt2 * 1e3, 4)
        out["value"] = round(n / dt2 / 1e6, 1)
        out["streams"] = "sdr_fm_set_overlap: audio FIR / de-emphasis / resampler of chunk k beside the channel filter of chunk k+1"
        tf2 = fps * n / dt2 / 1e12
        out["roofline"].update({"kernel_ms_one_stream": out["roofline"]["kernel_ms"], "kernel_ms": round(dt2 * 1e3, 4),
                                "achieved": round(tf2, 1), "frac": round(tf2 / FP32_PEAK_TFLOPS, 4),
                                "achieved_one_stream": round(tf, 1), "frac_one_stream": round(tf / FP32_PEAK_TFLOPS, 4),
                                "hbm_achieved_gbs": round(bps * n / dt2 / 1e9, 1), "hbm_frac": round(bps * n / dt2 / 1e9 / HBM_PEAK_GBS, 4)})
    if not skip_cpu:
        from oracle import dsp as odsp
        want, reps, tc = timed_repeat(lambda: odsp.fm_radio_chain(iq, samp_rate=fs, shift=200e3), 2.0)
        out["cpu_baseline"] = {"value": round(reps * m / tc / 1e6, 2), "unit": "Msps", "cores": 1, "kind": "port",
                               "sample": "2^22 samples x %d passes, oracle C blocks, %.1f s" % (reps, tc)}
        r.reset()
        half = m // 2 + 12345
        r.process(iq[:half])                      # two ragged chunks of the stream
        got = [r.fetch(4)]
        r.process(iq[half:])
        got.append(r.fetch(4))
        got = np.concatenate(got)
        err = float(np.max(np.abs(got - want[4]))) / float(np.max(np.abs(want[4]))) if got.shape == want[4].shape else 1.0
        if got.shape != want[4].shape or err > 1e-5:
            raise SystemExit("FM audio differs from the oracle: rel err %g" % err)
        out["verified"] = "audio of a 2^22-sample capture fed in two ragged chunks within 1e-5 of the oracle (rel err %.1e)" % err
    r.close()
    return out




def r_shape_workload(torch, dev, skip_cpu, ncap=64, n=1 << 19, fs=2e6, steps=20):
    """Secondary line: the reference-faithful "R" shape of SURVEY 8d-1 = TS33C/live_decoder.grc
    as wired (samp_rate 2e6, low_pass_filter 20k/40k Hamming = 121 taps at decimation 1,
    mag^2, threshold .05/.15, keep_one_in_n(100), decoder at 20 kHz).  484 flop per input
    sample: fp32-FMA bound, runs on the f32 matrix cores (k_frontend_mfma)."""
    from sdr_amd import chain, _lib
    iq, pkts = make_batch(torch, dev, ncap, seed=33, FS=fs, N_PER_CAPTURE=n, F_OFF=0.0)
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    c = chain.Chain(fs, decim=1, taps=taps, keep_n=100, decoder=chain.DEC_TS33C, max_batch=ncap, max_frames=4,
                    max_samples=n)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    prewarm(torch, lambda: c.batch_process(iq))
    for _ in range(3):
        c.batch_process(iq)
    torch.cuda.synchronize()
    c.profile_enable(True, PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        c.batch_process(iq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    front_ms, back_ms = c.profile_read()
    got = {}
    for f in c.fetch_frames():
        if f.status == _lib.FRAME_OK:
            got.setdefault(f.capture, []).append(f.packet())
    n_ok = sum(1 for i in range(ncap) if got.get(i) == [pkts[i], pkts[i]])
    if n_ok != ncap:
        raise SystemExit("R shape: only %d of %d captures decoded to the transmitted packets" % (n_ok, ncap))
    fk = float(np.mean(front_ms))
    flops = 4.0 * taps.size * ncap * n
    r_traffic, r_traffic_src = pmc_traffic("k_frontend_win r_shape", "captures=%d" % ncap)
    out = {"value": round(ncap * n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4),
           "captures": ncap, "samples_per_capture": n, "sample_rate": fs, "ntaps": int(taps.size), "decim": 1,
           "keep_n": 100, "chain": "fir_filter_ccf->mag2->threshold->slicer->keep_one_in_n->ts33c_decoder",
           "roofline": {"bound": "mfma", "kernel": "k_frontend_win", "achieved": round(flops / (fk * 1e-3) / 1e12, 1),
                        "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flops / (fk * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                        "traffic": r_traffic, "traffic_source": r_traffic_src,
                        "kernel_ms": round(fk, 4), "backend_kernel_ms": round(float(np.mean(back_ms)), 4),
                        "flops_per_launch": flops, "note": "algorithmic flops = 4*ntaps per input sample (real taps); peak = the f32 "
                        "FMA rate (packed VALU = f32-input MFMA, 157.3 TF); the sliding-window VALU kernel issues exactly "
                        "the algorithmic fmas (the banded-Toeplitz MFMA form, SDR_FRONTEND=mfma, (15+ntaps)/ntaps times as many)"},
           "verified": "all captures decode to the transmitted packets"}
    overlapped_pass(torch, c, lambda: c.batch_process(iq), steps, out, ncap * n)
    if not skip_cpu:
        from oracle import decoders as odec
        from oracle import dsp as odsp
        bits = c.fetch_bits().reshape(ncap, -1)
        x = iq[0].cpu().numpy().view(np.complex64).reshape(-1)
        def cpu_pass():
            w = odsp.ook_chain(x, taps, 1, 1.0, 0.05, 0.15, 100)
            dd = odec.TS33CStreamDecoder(sample_rate=20000)
            dd.work([w])
            return w, dd
        (want, d), reps, tc = timed_repeat(cpu_pass, 2.0)
        if not np.array_equal(bits[0], want) or [f["packet"] for f in d.frames] != got.get(0):
            raise SystemExit("R shape: capture 0 differs from the oracle")
        out["cpu_baseline"] = {"value": round(reps * n / tc / 1e6, 2), "unit": "Msps", "cores": 1, "kind": "port",
                               "sample": "capture 0 (2^19 samples) x %d passes, oracle C chain + numpy TS33C decoder, %.2f s" % (reps, tc)}
        out["verified"] += "; capture 0 bit-exact vs oracle"
    c.close()
    return out


def cu8_workload(torch, dev, iq, pkts, taps, steps=20):
    """Secondary line: the headline batch quantised to rtl-sdr bytes (2 B per
    sample instead of 8), converted while staged.  Same chain, same decoder."""
    from sdr_amd import chain, _lib
    ncap, n = iq.shape[0], iq.shape[1]
    u8 = torch.empty((ncap, n, 2), dtype=torch.uint8, device=dev)
    for a in range(0, ncap, 64):
        u8[a:a + 64] = torch.clamp(torch.round(iq[a:a + 64] * 128.0 + 127.4), 0, 255).to(torch.uint8)
    c = chain.Chain(FS, decim=DECIM, center_freq=F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_batch=ncap,
                    max_frames=4, max_samples=n, input_format="cu8")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    prewarm(torch, lambda: c.batch_process(u8))
    for _ in range(3):
        c.batch_process(u8)
    torch.cuda.synchronize()
    c.profile_enable(True, PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        c.batch_process(u8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    front_ms, back_ms = c.profile_read()
    got = {}
    for f in c.fetch_frames():
        if f.status == _lib.FRAME_OK:
            got.setdefault(f.capture, []).append(f.packet())
    n_ok = sum(1 for i in range(ncap) if got.get(i) == [pkts[i], pkts[i]])
    if n_ok != ncap:
        raise SystemExit("cu8: only %d of %d captures decoded to the transmitted packets" % (n_ok, ncap))
    fk = float(np.mean(front_ms))
    bps = 2.0 + 1.0 / DECIM
    ach = bps * ncap * n / (fk * 1e-3) / 1e9
    c.close()
    return {"value": round(ncap * n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4),
            "captures": ncap, "samples_per_capture": n, "input": "cu8 (u8 I, u8 Q), (x-127.4)/128 while staging",
            "roofline": {"bound": "hbm", "kernel": "k_frontend_raw8", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         **traffic_fields("k_frontend_raw8 cu8", "captures_per_gpu=%d" % ncap),
                         "kernel_ms": round(fk, 4), "backend_kernel_ms": round(float(np.mean(back_ms)), 4),
                         "bytes_per_launch": bps * ncap * n,
                         "note": "2.02 algorithmic bytes per sample: the kernel is latency/issue bound, not HBM "
                                 "bound -- the figure of merit is the sample rate next to the complex64 headline"},
            "verified": "all captures decode to the transmitted packets"}


def tfa_fsk_workload(torch, dev, skip_cpu, ncap=64, n=1 << 20, fs=2e6, dev_hz=60e3, steps=20):
    """Secondary line: BASELINE.json configs[2] / SURVEY 8d-3 -- TFA 30.3180.IT 2-level FSK at 38400 Bd:
    LPF (27 taps, TFA/decoder.grc:1163-1234) at decimation 1 -> quadrature_demod (TFA/decoder.grc:450-481)
    -> threshold +-0.2 -> slicer -> TFA stream decoder at 2 MS/s (symbol length from the sync pairs,
    CRC-8).  One sliced byte per input sample: 9 algorithmic bytes per sample."""
    from sdr_amd import chain, synth, _lib
    rng = np.random.default_rng(30)
    nenv = 8
    envs, pkts = np.zeros((nenv, n), np.float32), []
    for e in range(nenv):
        sym, pk = [], []
        for _ in range(18):
            p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10,
                                 humidity=int(rng.integers(0, 100)))
            pk.append(p)
            sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=20000.0 + 2000.0 * rng.random())
        env = synth.envelope(sym, fs, lead_in_us=800.0)
        if env.size > n - 2500:
            raise SystemExit("tfa workload: envelope does not fit")
        envs[e, :env.size] = env
        pkts.append(pk)
    iq = torch.empty((ncap, n, 2), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(30)
    for i in range(ncap):
        f = torch.from_numpy(envs[i % nenv]).to(dev).to(torch.float64) * 2.0 - 1.0
        ph = torch.cumsum(f, 0) * (2 * np.pi * dev_hz / fs) + float(rng.random() * 2 * np.pi)
        iq[i, :, 0] = (0.8 * torch.cos(ph)).to(torch.float32)
        iq[i, :, 1] = (0.8 * torch.sin(ph)).to(torch.float32)
        iq[i] += 0.01 * torch.randn((n, 2), generator=g, device=dev, dtype=torch.float32)
    taps = chain.low_pass(1, fs, 500e3, 250e3, chain.WIN_BLACKMAN)
    gain = np.float32(fs / (2 * np.pi * dev_hz))
    c = chain.Chain(fs, decim=1, taps=taps, demod=chain.DEMOD_QUAD, demod_gain=gain, thr=(-0.2, 0.2),
                    decoder=chain.DEC_TFA_GRC, max_batch=ncap, max_frames=32, max_samples=n)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    prewarm(torch, lambda: c.batch_process(iq))
    for _ in range(3):
        c.batch_process(iq)
    torch.cuda.synchronize()
    c.profile_enable(True, PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        c.batch_process(iq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    front_ms, back_ms = c.profile_read()
    frames = c.fetch_frames()
    n_ok = 0
    for i in range(ncap):
        n_ok += [f.packet() for f in frames if f.capture == i and f.status == _lib.FRAME_OK] == pkts[i % nenv]
    if n_ok != ncap:
        raise SystemExit("TFA FSK: only %d of %d captures decoded to the transmitted packets" % (n_ok, ncap))
    fk = float(np.mean(front_ms))
    bps = 9.0
    ach = bps * ncap * n / (fk * 1e-3) / 1e9
    out = {"value": round(ncap * n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4), "captures": ncap,
           "samples_per_capture": n, "sample_rate": fs, "ntaps": int(taps.size), "decim": 1, "packets_per_capture": 18,
           "chain": "fir_filter_ccf->quadrature_demod->threshold->slicer->tfa_stream_decoder",
           "roofline": {"bound": "hbm", "kernel": "k_frontend_win", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        **traffic_fields("k_frontend_win tfa", "captures=%d" % ncap), "kernel_ms": round(fk, 4),
                        "backend_kernel_ms": round(float(np.mean(back_ms)), 4), "bytes_per_launch": bps * ncap * n,
                        "note": "27 taps at decimation 1 + one fast_atan2f per sample: 108 + ~90 flop-equivalents per 9 bytes, "
                                "bound by VALU instruction issue (profiles/r02_a_win_sq.csv), not by HBM"},
           "verified": "all captures decode to the transmitted packets (CRC-8 ok)"}
    overlapped_pass(torch, c, lambda: c.batch_process(iq), steps, out, ncap * n)
    if not skip_cpu:
        from oracle import decoders as odec
        from oracle import dsp as odsp
        bits = c.fetch_bits().reshape(ncap, -1)
        x = iq[0].cpu().numpy().view(np.complex64).reshape(-1)
        def cpu_pass():
            w = odsp.quad_demod(odsp.fir_ccf(x, taps, 1), gain)
            b, _, _ = odsp.slice_bits(w, 1.0, -0.2, 0.2, 1)
            oo = odec.TFAStreamDecoder(sample_rate=int(fs), nan_guard=True)
            oo.work([b])
            return b, oo
        (wb, o), reps, tc = timed_repeat(cpu_pass, 2.0)
        mine = [f for f in frames if f.capture == 0]
        if not np.array_equal(bits[0], wb) or len(mine) != len(o.frames) or \
                [f["packet"] for f in o.frames if f["status"] == "ok"] != [f.packet() for f in mine if f.status == 0]:
            raise SystemExit("TFA FSK: capture 0 differs from the oracle")
        out["cpu_baseline"] = {"value": round(reps * n / tc / 1e6, 2), "unit": "Msps", "cores": 1, "kind": "port",
                               "sample": "capture 0 (2^20 samples) x %d passes, oracle C blocks + numpy TFA decoder, %.2f s" % (reps, tc)}
        out["verified"] += "; capture 0 bit-exact vs oracle (sliced bytes, frames)"
    c.close()
    return out


def make_mixed_batch(torch, dev, ncap, n=N_PER_CAPTURE, fs=FS, seed=4242):
    """configs[3] as worded: one somfy RTS / berner / mumbi emitter per capture (timings of the reference's
    transmitter.py scripts, random payloads), carrier + AWGN applied on the GPU.  -> (iq, [(decoder, payload)])"""
    from sdr_amd import chain, synth
    rng = np.random.default_rng(seed)
    env = np.zeros((ncap, n), np.uint8)
    want = []
    codes = sorted(synth.MUMBI_CODES)
    for i in range(ncap):
        kind = i % 3
        if kind == 0:
            fr = synth.somfy_frame(0xA0, int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 24)))
            sym, lead = synth.somfy_symbols(fr, repeats=0), 3000.0
            want.append((chain.DEC_SOMFY, fr))
        elif kind == 1:
            code = int(rng.integers(0, 1 << 32))
            sym, lead = synth.berner_symbols(code, repeats=5), 3000.0
            want.append((chain.DEC_BERNER, code.to_bytes(4, "big")))
        else:
            code = bytes(bytearray(synth.MUMBI_CODES[codes[int(rng.integers(0, len(codes)))]]))
            sym, lead = synth.mumbi_symbols(code, repeats=4), 9000.0
            want.append((chain.DEC_MUMBI, bytes(bytearray([code[0], code[1], code[2], code[3], code[4] & 0xC0]))))
        env[i] = synth.envelope(sym, fs, lead_in_us=lead + 200.0 * rng.random(), total=n)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    iq = torch.empty((ncap, n, 2), dtype=torch.float32, device=dev)
    t = torch.arange(n, dtype=torch.float64, device=dev)
    ph0 = torch.from_numpy(rng.random(ncap) * 2 * np.pi).to(dev)
    for a in range(0, ncap, 64):
        b = min(ncap, a + 64)
        e = torch.from_numpy(env[a:b]).to(dev).to(torch.float32) * 0.7
        ph = (2 * np.pi * F_OFF / fs) * t[None, :] + ph0[a:b, None]
        iq[a:b, :, 0] = e * torch.cos(ph).to(torch.float32)
        iq[a:b, :, 1] = e * torch.sin(ph).to(torch.float32)
        iq[a:b] += 0.02 * torch.randn((b - a, n, 2), generator=g, device=dev, dtype=torch.float32)
    return iq, want


def mixed_emitters_workload(torch, dev, skip_cpu, ncap=1250, n=N_PER_CAPTURE, fs=FS, steps=10):
    """Secondary line: BASELINE.json configs[3] as worded -- a batch of OOK/PWM captures from somfy RTS,
    berner and mumbi transmitters (one emitter per capture, random payloads; timings of the reference's
    transmitter.py scripts), channel filter at decimation 10 (100 kS/s sliced stream: the pigpio rules
    need ~10 us edge resolution), then ALL THREE rule decoders over every capture (sdr_redecode).
    The same batch is checked at full size in tests/test_gpu_fullsize.py."""
    from sdr_amd import chain, synth, _lib
    D = 10
    iq, want = make_mixed_batch(torch, dev, ncap, n, fs)
    taps = chain.low_pass(1.0, fs, 20e3, 40e3)
    c = chain.Chain(fs, decim=D, center_freq=F_OFF, taps=taps, decoder=chain.DEC_SOMFY, max_batch=ncap, max_frames=8,
                    max_samples=n)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    decs = (chain.DEC_SOMFY, chain.DEC_BERNER, chain.DEC_MUMBI)

    def step(collect=None):
        c.batch_process(iq)
        if collect is not None:
            collect[chain.DEC_SOMFY] = c.fetch_frames()
        for d in decs[1:]:
            c.redecode(d)
            if collect is not None:
                collect[d] = c.fetch_frames()

    prewarm(torch, step)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    c.profile_enable(True, PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    front_ms, back_ms = c.profile_read()
    c.profile_enable(False)
    got = {}
    step(got)
    n_ok = 0
    for i, (d, payload) in enumerate(want):
        mine = [f for f in got[d] if f.capture == i]
        if d == chain.DEC_SOMFY:
            ok = len(mine) >= 1 and all(bytes(bytearray(f.raw[:7])) == payload and f.status == _lib.FRAME_OK for f in mine)
        else:
            ok = len(mine) >= 1 and all(f.packet() == payload for f in mine)
        n_ok += ok
    if n_ok != ncap:
        raise SystemExit("mixed emitters: only %d of %d captures decoded to the transmitted payloads" % (n_ok, ncap))
    # profile slots come three to a step: (front, somfy back end), then one per redecode whose "front" gap is empty
    fk = float(np.mean(front_ms[0::3]))
    bk = 3.0 * float(np.mean(back_ms))
    bps = 8.0 + 1.0 / D
    ach = bps * ncap * n / (fk * 1e-3) / 1e9
    traffic, _src = pmc_traffic("k_frontend<true, false> fc32 D=10", "ook_batch_somfy_berner_mumbi captures=%d" % ncap)
    out = {"value": round(ncap * n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4), "captures": ncap,
           "samples_per_capture": n, "sample_rate": fs, "ntaps": int(taps.size), "decim": D,
           "emitters": "somfy RTS / berner / mumbi, one per capture", "decoders_per_capture": 3,
           "chain": "xlating_fir->mag2->threshold->slicer->{somfy, berner, mumbi} rule decoders",
           "roofline": {"bound": "hbm", "kernel": "k_frontend<true, false>", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": _src, "kernel_ms": round(fk, 4),
                        "backend_kernel_ms": round(bk, 4), "bytes_per_launch": bps * ncap * n},
           "verified": "every capture decodes to its transmitted payload under its own decoder"}
    if not skip_cpu:
        from oracle import decoders as odec
        from oracle import dsp as odsp
        bits = c.fetch_bits().reshape(ncap, -1)
        sniff = {chain.DEC_SOMFY: (odec.SomfySniffer, 20000), chain.DEC_BERNER: (odec.BernerSniffer, None),
                 chain.DEC_MUMBI: (odec.MumbiSniffer, 8000)}
        xs = [iq[i].cpu().numpy().view(np.complex64).reshape(-1) for i in (0, 1, 2)]

        def cpu_pass():
            res = []
            for i in (0, 1, 2):
                wb = odsp.ook_chain(xs[i], taps, D, 1.0, 0.05, 0.15, 1, fc=F_OFF, fs=fs)
                cls, wd = sniff[want[i][0]]
                sn = cls()
                for lvl, tick in odec.edges_to_events(wb, int(fs / D), watchdog_us=wd):
                    sn.cbf(lvl, tick)
                res.append((wb, sn))
            return res
        res, reps, tc = timed_repeat(cpu_pass, 2.0)
        for i, (wb, sn) in enumerate(res):
            if not np.array_equal(bits[i], wb):
                raise SystemExit("mixed emitters: capture %d sliced bytes differ from the oracle" % i)
            mine = [f for f in got[want[i][0]] if f.capture == i]
            ref = [fr["raw"] if isinstance(fr, dict) else fr for fr in sn.frames]
            have = [bytes(bytearray(f.raw[:7])) if want[i][0] == chain.DEC_SOMFY else f.packet() for f in mine]
            if ref != have:
                raise SystemExit("mixed emitters: capture %d frames differ from the oracle" % i)
        out["cpu_baseline"] = {"value": round(reps * 3 * n / tc / 1e6, 2), "unit": "Msps", "cores": 1, "kind": "port",
                               "sample": "captures 0-2 (one per emitter) x %d passes, oracle C chain + rule restatements, %.2f s" % (reps, tc)}
        out["verified"] += "; captures 0-2 bit-exact vs oracle (sliced bytes and frames)"
    c.close()
    return out


def make_wideband(torch, dev, n=1 << 26, fs=20e6, M=64):
    """One wideband capture: noise + a TS33C, a berner and a mumbi transmitter in three of the 64 sub-bands.
    -> (iq [n, 2] on the GPU, dict of what was transmitted where)"""
    from sdr_amd import synth
    rng = np.random.default_rng(64)
    pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
    sym = []
    for _ in range(3):
        sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
    ch, ch_berner, ch_mumbi = 5, 40, 17
    berner_code = int(rng.integers(0, 1 << 32))
    codes = sorted(synth.MUMBI_CODES)
    mumbi_code = bytes(bytearray(synth.MUMBI_CODES[codes[int(rng.integers(0, len(codes)))]]))
    mumbi_want = bytes(bytearray([mumbi_code[0], mumbi_code[1], mumbi_code[2], mumbi_code[3], mumbi_code[4] & 0xC0]))
    emitters = [(ch, sym, 5000.0), (ch_berner, synth.berner_symbols(berner_code, repeats=5), 150000.0),
                (ch_mumbi, synth.mumbi_symbols(mumbi_code, repeats=8), 400000.0)]
    g = torch.Generator(device=dev)
    g.manual_seed(64)
    iq = 0.02 * torch.randn((n, 2), generator=g, device=dev, dtype=torch.float32)
    step_n = 1 << 22
    for chan, esym, lead in emitters:
        env = torch.from_numpy(synth.envelope(esym, fs, lead_in_us=lead, total=n)).to(dev)
        for a in range(0, n, step_n):
            t = torch.arange(a, a + step_n, dtype=torch.float64, device=dev)
            ph = (2 * np.pi * chan / M) * t
            e = env[a:a + step_n].to(torch.float32) * 0.5
            iq[a:a + step_n, 0] += e * torch.cos(ph).to(torch.float32)
            iq[a:a + step_n, 1] += e * torch.sin(ph).to(torch.float32)
        del env
    return iq, {"ts33c": (ch, pkt), "berner": (ch_berner, berner_code), "mumbi": (ch_mumbi, mumbi_want)}


def pfb_workload(torch, dev, skip_cpu, n=1 << 26, fs=20e6, steps=10):
    """Secondary line (BASELINE.json configs[4]): one 2^26-sample wideband capture at 20 Msps
    -> 64 sub-bands of 312.5 kHz (481-tap prototype, 8 per branch) -> mag^2 -> threshold ->
    slicer -> keep_one_in_n(16) -> the TS33C stream decoder and the somfy / berner / mumbi rule decoders
    over every channel (sdr_redecode_multi); a TS33C, a berner and a mumbi transmitter sit in three sub-bands."""
    from sdr_amd import chain, synth, _lib
    M, keep = 64, 16
    rate = fs / M / keep
    iq, sent = make_wideband(torch, dev, n, fs, M)
    (ch, pkt), (ch_berner, berner_code), (ch_mumbi, mumbi_want) = sent["ts33c"], sent["berner"], sent["mumbi"]
    taps = chain.low_pass(1.0, fs, 100e3, 100e3)
    c = chain.Chain(fs, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=n,
                    max_frames=64, channels=M)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    x = iq.reshape(1, n, 2)
    rules = (chain.DEC_SOMFY, chain.DEC_BERNER, chain.DEC_MUMBI)

    def step(collect=None):
        c.batch_process(x)                      # channelizer + TS33C stream decoder on the 64 rows
        if collect is not None:
            collect[chain.DEC_TS33C] = c.fetch_frames()
        c.redecode_multi(rules, rate)           # the three rule decoders over the same sliced rows, one launch
        if collect is not None:
            for k, d in enumerate(rules):
                collect[d] = c.fetch_frames_multi(k)

    prewarm(torch, step)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    c.profile_enable(True, PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    front_ms, back_ms = c.profile_read()
    c.profile_enable(False)
    per = 2                                     # profile slots per step: channelizer pass + the multi-decoder launch
    front_ms, back_ms = front_ms[0::per], [sum(back_ms[i:i + per]) for i in range(0, len(back_ms) - per + 1, per)]
    got = {}
    step(got)
    good = [f for f in got[chain.DEC_TS33C] if f.status == _lib.FRAME_OK]
    fb, fm = got[chain.DEC_BERNER], got[chain.DEC_MUMBI]
    if not fb or any(f.capture != ch_berner for f in fb) or fb[0].packet() != berner_code.to_bytes(4, "big"):
        raise SystemExit("channelizer: berner frames %s" % [(f.capture, f.packet().hex()) for f in fb])
    if not fm or any(f.capture != ch_mumbi for f in fm) or not any(f.packet() == mumbi_want for f in fm):
        raise SystemExit("channelizer: mumbi frames %s" % [(f.capture, f.packet().hex()) for f in fm])
    if got[chain.DEC_SOMFY]:
        raise SystemExit("channelizer: somfy frames without a somfy transmitter")
    if [f.capture for f in good] != [ch] * len(good) or [f.packet() for f in good] != [pkt] * 3:
        raise SystemExit("channelizer: expected the 3 transmitted frames in channel %d, got %s"
                         % (ch, [(f.capture, f.packet().hex()) for f in good]))
    fk = float(np.mean(front_ms))
    bps = 8.0 + 1.0 / keep
    ach = bps * n / (fk * 1e-3) / 1e9
    out = {"value": round(n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4), "samples": n,
           "sample_rate": fs, "channels": M, "ntaps": int(taps.size), "keep_n": keep,
           "chain": "pfb64->mag2->threshold->slicer->keep_one_in_n->{ts33c, somfy, berner, mumbi} decoders per channel",
           "decoder_passes_per_step": 4, "decoder_launches_per_step": 2,
           "roofline": {"bound": "hbm", "kernel": "k_pfb<8>", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        **traffic_fields("k_pfb<8>", "samples=2^%d" % int(np.log2(n))),
                        "kernel_ms": round(fk, 4), "backend_kernel_ms": round(float(np.mean(back_ms)), 4),
                        "bytes_per_launch": bps * n},
           "verified": "the 3 TS33C frames decode in channel 5, the berner code in channel 40, the mumbi code in "
                       "channel 17, nothing valid anywhere else"}
    overlapped_pass(torch, c, step, steps, out, n, per=2)
    if not skip_cpu:
        from oracle import dsp as odsp
        m = 1 << 21
        xs = iq[:m].cpu().numpy().view(np.complex64).reshape(-1)
        bits = c.fetch_bits().reshape(M, -1)
        def cpu_pass():
            y = odsp.pfb_channelize(xs, taps, M)
            return np.stack([odsp.slice_bits(odsp.mag2(y[k]), 1.0, 0.05, 0.15, keep)[0] for k in range(M)])
        want, reps, tc = timed_repeat(cpu_pass, 2.0)
        if not np.array_equal(bits[:, :want.shape[1]], want):
            raise SystemExit("channelizer: sliced bytes of the first 2^21 samples differ from the oracle")
        out["cpu_baseline"] = {"value": round(reps * m / tc / 1e6, 2), "unit": "Msps", "cores": 1, "kind": "port",
                               "sample": "first 2^21 samples x %d passes, oracle C channelizer + slicer, %.2f s" % (reps, tc)}
        out["verified"] += "; all 64 sliced streams of the first 2^21 samples bit-exact vs oracle"
    c.close()
    return out


def pfb_segments_workload(torch, dist, dev, rank, world, n=1 << 26, fs=20e6, steps=5):
    """BASELINE.json configs[4] as worded -- "20 Msps capture channelized into 64 sub-bands, all sensor decoders per
    channel on 8 GPUs" -- for ONE capture: rank r takes the r-th contiguous segment (boundaries multiples of
    64 * keep_n) with the channelizer's FIR-history halo in front of it (sdr_seek), channelizes and slices it; the
    per-row threshold summaries and the sliced rows (1/(64*16) byte per input sample and channel) are gathered
    over RCCL (segments.gather_rows: two all_gathers); the 64 stitched rows are dealt out to the ranks and decoded
    (sdr_batch_process_bits), and the frame records are gathered as in the headline.  Every step does all of
    that; strong scaling (the capture is fixed, the segment shrinks with the rank count).  Verified on rank 0
    against ONE sdr_batch_process pass over the whole capture: same stitched bytes, same frames."""
    from sdr_amd import chain, segments, _lib
    from sdr_amd import dist as sdist
    M, keep, max_frames = 64, 16, 64
    rate = fs / M / keep
    iq, sent = make_wideband(torch, dev, n, fs, M)      # same seed on every rank: every rank holds the capture
    taps = chain.low_pass(1.0, fs, 100e3, 100e3)
    a, b = segments.segment_bounds(n, world, M * keep)[rank]
    front = chain.Chain(fs, taps=taps, keep_n=keep, decoder=chain.DEC_NONE, max_samples=max(b - a, 1), channels=M)
    dec = chain.Chain(fs, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=n,
                      max_frames=max_frames, channels=M)
    for c in (front, dec):
        c.set_stream(torch.cuda.current_stream().cuda_stream)
    H = front.history_len()
    per = -(-M // world)
    r0, r1 = min(M, rank * per), min(M, (rank + 1) * per)
    rec = torch.zeros(per * max_frames * sdist.FRAME_BYTES, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(per, dtype=torch.int32, device=dev)
    res = {}

    def step():
        lo = max(0, a - H)
        front.seek(a, iq[lo:a] if a > lo else None)
        front.process_iq(iq[a:b])
        infos = front.segment_info_rows()
        nb = int(infos[0].n_bits)
        bits = torch.zeros((M, max(nb, 1)), dtype=torch.uint8, device=dev)
        front.export_bits(bits)
        stream, _off = segments.gather_rows(bits, nb, [i.n_undetermined for i in infos], [i.has_event for i in infos],
                                            [i.last_state for i in infos], world)
        rec.zero_()
        cnt.zero_()
        if r1 > r0:
            dec.batch_process_bits(stream[r0:r1])
            dec.export_frames(rec, cnt)
        res["stream"] = stream
        res["gathered"] = sdist.gather_frames(rec, cnt, world)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item()) / steps
    out = None
    if rank == 0:
        all_rec, all_cnt = res["gathered"]
        got = []
        recs = np.ascontiguousarray(all_rec.cpu().numpy())
        cnts = all_cnt.cpu().numpy()
        for g in range(cnts.size):
            r, i = divmod(g, per)
            row = r * per + i
            for k in range(int(cnts[g])):
                off = (g * max_frames + k) * sdist.FRAME_BYTES
                f = _lib.Frame.from_buffer_copy(recs[off:off + sdist.FRAME_BYTES].tobytes())
                got.append((row, f.status, f.packet(), int(f.end_pos)))
        whole = chain.Chain(fs, taps=taps, keep_n=keep, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=n,
                            max_frames=max_frames, channels=M)
        whole.batch_process(iq.reshape(1, n, 2))
        want = [(f.capture, f.status, f.packet(), int(f.end_pos)) for f in whole.fetch_frames()]
        want_bits = whole.fetch_bits().reshape(M, -1)
        whole.close()
        if not np.array_equal(res["stream"].cpu().numpy().view(np.int8), want_bits):
            raise SystemExit("pfb segments: stitched rows differ from the single pass")
        if got != want:
            raise SystemExit("pfb segments: frames differ from the single pass: %s vs %s" % (got[:4], want[:4]))
        ch, pkt = sent["ts33c"]
        good = [g_ for g_ in got if g_[1] == _lib.FRAME_OK]
        if [g_[0] for g_ in good] != [ch] * 3 or [g_[2] for g_ in good] != [pkt] * 3:
            raise SystemExit("pfb segments: expected the 3 transmitted TS33C frames in channel %d" % ch)
        out = {"value": round(n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4), "samples": n,
               "sample_rate": fs, "channels": M, "n_gpus": world, "scaling": "strong", "segment_samples": b - a,
               "halo_samples": H, "keep_n": keep,
               "chain": "per rank: sdr_seek(halo) -> pfb64 -> mag2 -> threshold -> slicer -> keep_one_in_n on its segment; "
                        "all_gather of per-row threshold summaries and sliced rows; rows dealt out -> ts33c decoder; all_gather of frame records",
               "collective_bytes_per_step": int(world * (3 * M + 1) * 8 + world * M * max(16, -(-((b - a) // (M * keep)) // 16) * 16)
                                                + world * (rec.numel() + 4 * cnt.numel())),
               "verified": "stitched rows and frame records equal ONE sdr_batch_process pass over the whole capture on rank 0; "
                           "the 3 transmitted TS33C frames decode in channel %d" % ch,
               "note": "host-orchestrated (seek, summaries read back, two all_gathers, decode, frame gather per step): "
                       "latency-bound, the figure of merit is that the segmented capture decodes to what one pass gives"}
    front.close()
    dec.close()
    del iq
    torch.cuda.empty_cache()
    return out


def long_capture_frames_workload(torch, dist, dev, rank, world, ncat=256, steps=5):
    """north_star's second multi-GPU case -- "each GPU gets a contiguous segment with FIR-history overlap ... RCCL is
    used only to gather decoded frames and counts" -- for ONE long TS33C capture (ncat headline captures back to
    back: 2^26 samples, 2 frames per 2^18): rank r runs the fused chain over its segment (sdr_seek with the halo),
    the 4-int64 threshold scan gives it its carry-in, it decodes its own bytes (sdr_segment_decode) and hands out its
    frames plus the run starts of the pieces that touch its ends (segments.gather_frames: one 16-KiB record and the
    kept frame records per rank); every rank decodes the joined table (sdr_process_runs).  No sliced byte leaves a
    GPU.  Strong scaling.  Verified on rank 0 against ONE streaming pass over the whole capture on one handle."""
    from sdr_amd import chain, segments
    n = ncat * N_PER_CAPTURE
    iq, _ = make_batch(torch, dev, ncat, seed=777)   # same seed on every rank: every rank holds the capture
    iq = iq.reshape(n, 2)
    taps = chain.low_pass(1.0, FS, 20e3, 40e3)
    a, b = segments.segment_bounds(n, world, DECIM)[rank]
    max_frames = 4 * ncat
    c = chain.Chain(FS, decim=DECIM, center_freq=F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_samples=max(b - a, 1),
                    max_frames=max_frames)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    H = c.history_len()
    res = {}

    def decode_runs(pos, fl, ll):
        c.seek(0)
        c.process_runs(pos, fl, ll)
        return c.fetch_frames()

    def step():
        lo = max(0, a - H)
        c.seek(a, iq[lo:a] if a > lo else None)
        c.process_iq(iq[a:b])
        info = c.segment_info()
        carries, _nb = segments.gather_carries(info.n_bits, info.n_undetermined, info.has_event, info.last_state, world,
                                               device=dev)
        pieces = c.segment_decode(carries[rank], 1024)
        frames = c.fetch_frames() if pieces.n_bits else []
        res["frames"], res["stats"] = segments.gather_frames(pieces, frames, world, decode_runs, cap=1024, device=dev)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item()) / steps
    c.close()
    out = None
    if rank == 0:
        if res["frames"] is None:
            raise SystemExit("long capture: a segment's head / tail table did not fit (flags), stats %s" % res["stats"])
        got = [(int(f.end_pos), f.status, f.packet()) for f in res["frames"]]
        whole = chain.Chain(FS, decim=DECIM, center_freq=F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_samples=n,
                            max_frames=max_frames)
        whole.process_iq(iq)
        want = [(int(f.end_pos), f.status, f.packet()) for f in whole.fetch_frames()]
        whole.close()
        if got != want or len(want) < 2 * ncat - 2:
            raise SystemExit("long capture: %d frames by segments, %d by the single pass; first difference %s"
                             % (len(got), len(want), next(((g, w) for g, w in zip(got, want) if g != w), None)))
        st = res["stats"]
        out = {"value": round(n / dt / 1e6, 1), "unit": "Msps", "ms_per_step": round(dt * 1e3, 4), "samples": n,
               "sample_rate": FS, "n_gpus": world, "scaling": "strong", "segment_samples": b - a, "halo_samples": H,
               "frames": len(got), "frames_decoded_locally": st["local_frames"], "frames_from_the_joined_table": st["stitched_frames"],
               "runs_in_the_joined_table": st["runs"], "collective_bytes_per_rank_and_step": 32 + st["payload_bytes"],
               "chain": "per rank: sdr_seek(halo) -> xlating_fir -> mag2 -> threshold -> slicer on its segment; all_gather of 4 int64 "
                        "(threshold scan); sdr_segment_decode; all_gather of one 16-KiB run-start record and of the kept frame "
                        "records; sdr_process_runs on the joined table",
               "verified": "frames (end position, status, packet) equal ONE streaming pass over the whole capture on rank 0",
               "note": "host-orchestrated and latency-bound (two read-backs and three small collectives per step); no sliced "
                       "byte leaves a GPU"}
    del iq
    torch.cuda.empty_cache()
    return out


PMC_FILE = "r03_k_pmc.json"   # the committed FETCH_SIZE / WRITE_SIZE passes of the current kernels (tools/profile_round.sh)


PREWARM_S = float(os.environ.get("SDR_BENCH_PREWARM", "0.15"))
# kernel durations inside the timed region come from HIP events around every PROFILE_EVERY-th processing call of a handle
# (its sdr_redecode* calls are timed with it): an event is a packet of its own on the queue, three per call were 10 us
# of the 0.44-ms headline step (tools/exp_prof_cost.py)
PROFILE_EVERY = int(os.environ.get("SDR_BENCH_PROFILE_EVERY", "3"))


def prewarm(torch, fn, seconds=None):
    """Untimed: the workload's own step for `seconds` of wall time before its warm-up steps.  A GPU that has been idle
    (or running the generators' small kernels) needs tens of milliseconds of load before its clocks settle; five
    0.5-ms steps do not get there (measured on one box: 0.470 ms per headline step after 5 warm-up steps, 0.448 after
    300 -- the latency-bound back end 0.0466 -> 0.0414 ms shows it is the clock).  SDR_BENCH_PREWARM=0 switches it off
    (the counter passes of tools/profile_round.sh)."""
    seconds = PREWARM_S if seconds is None else seconds
    if seconds <= 0:
        return
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()


def progress(msg):
    """one line per workload on stderr: a long run (under a profiler, say) shows where it is"""
    print("[bench] " + msg, file=sys.stderr, flush=True)


def overlapped_pass(torch, c, step, steps, out, units, per=1, warm=3):
    """The same steps again with sdr_set_overlap on: ms_per_step / value of `out` become the overlapped ones, the
    one-stream step and the front-end kernel's time beside a running back end are kept next to them."""
    if not OVERLAP:
        return
    progress("  second pass, overlapped")
    c.set_overlap(True)
    prewarm(torch, step)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    c.profile_enable(True, PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    front_ms, _back = c.profile_read()
    c.profile_enable(False)
    c.set_overlap(False)
    out["ms_per_step_one_stream"] = out["ms_per_step"]
    out["value"] = round(units / dt / 1e6, 1)
    out["ms_per_step"] = round(dt * 1e3, 4)
    out["streams"] = "sdr_set_overlap: back end of step k beside the front end of step k+1"
    out["roofline"]["kernel_ms_overlapped"] = round(float(np.mean(front_ms[0::per])), 4)


_PMC_CACHE = {}


def _pmc_file():
    """profiles/<PMC_FILE> once per process, plus whether the kernel sources are still the ones the passes ran on"""
    if "rec" not in _PMC_CACHE:
        root = os.path.dirname(os.path.abspath(__file__))
        try:
            with open(os.path.join(root, "profiles", PMC_FILE)) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            rec = {}
        fresh = None
        if rec.get("csrc_sha256"):
            import hashlib
            h = hashlib.sha256()
            d = os.path.join(root, "sdr_amd", "csrc")
            for name in sorted(os.listdir(d)):
                if name.endswith((".h", ".hip")):
                    h.update(name.encode())
                    h.update(open(os.path.join(d, name), "rb").read())
            fresh = h.hexdigest() == rec["csrc_sha256"]
        _PMC_CACHE["rec"], _PMC_CACHE["fresh"] = rec, fresh
    return _PMC_CACHE["rec"], _PMC_CACHE["fresh"]


def traffic_fields(kernel_key, workload_tag):
    t, src = pmc_traffic(kernel_key, workload_tag)
    return {"traffic": t, "traffic_source": src}


def pmc_traffic(kernel_key, workload_tag):
    """HBM bytes per launch of one kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE over this same command; counters cannot be read from inside the process).  None if the
    committed workload differs from the one being run.  The source string says whether the kernel sources
    (sdr_amd/csrc) are byte for byte the ones the passes ran on -- if not, the figure describes an older kernel."""
    rec, fresh = _pmc_file()
    try:
        k = rec["kernels"][kernel_key]
    except KeyError:
        return None, None
    if workload_tag not in k.get("workload", ""):
        return None, None
    src = "profiles/" + PMC_FILE
    if fresh is True:
        src += " (kernel sources unchanged since those passes)"
    elif fresh is False:
        src += " (STALE: sdr_amd/csrc changed since those passes)"
    return int(k["hbm_bytes"]), src


def launch_ranks(n, argv, rehearsal):
    """`python3 bench.py --gpus N` outside torchrun: start N fresh child processes, one per GPU, with the
    torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), BEFORE this process makes any GPU
    call (torch.cuda.device_count() does not initialise the runtime).  Rank 0 prints the JSON line on the
    inherited stdout.  Returns the exit code: non-zero if any rank failed (the others are then stopped)."""
    import socket
    import subprocess
    if not rehearsal:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py --gpus %d needs %d GPUs, this machine shows %d\n" % (n, n, have))
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (procs.index(p), code))
                for q in alive:
                    q.terminate()
        time.sleep(0.05)
    return rc


def stage_collectives_through_host(torch, dist):
    """SDR_BENCH_REHEARSAL=gloo-gpu only: gloo moves host tensors, so the collectives this file and sdr_amd use
    (all_gather_into_tensor, all_reduce) copy a GPU tensor to the host, run there and copy the result back."""
    ag, ar = dist.all_gather_into_tensor, dist.all_reduce

    def all_gather_into_tensor(out, inp, *a, **k):
        if not inp.is_cuda:
            return ag(out, inp, *a, **k)
        o = torch.empty(out.shape, dtype=out.dtype)
        ag(o, inp.cpu(), *a, **k)
        out.copy_(o)

    def all_reduce(t, *a, **k):
        if not t.is_cuda:
            return ar(t, *a, **k)
        h = t.cpu()
        ar(h, *a, **k)
        t.copy_(h)

    dist.all_gather_into_tensor = all_gather_into_tensor
    dist.all_reduce = all_reduce


def _oracle_low_pass():
    """the headline's channel filter from the oracle's firdes (the rehearsal has no GPU library to ask)"""
    from oracle import dsp as odsp
    return odsp.low_pass(1.0, FS, 20e3, 40e3)


def rehearsal_main(args, rank, world):
    """SDR_BENCH_REHEARSAL=gloo: the multi-rank skeleton of this file (rendezvous, frame-table gather, barriers,
    max over ranks, rank 0's line) on CPU ranks with made-up frame tables and NO chain work -- what
    tests/test_bench_launcher.py drives.  It measures nothing: value is null and the line says so."""
    import torch
    import torch.distributed as dist
    from sdr_amd import dist as sdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ncap, max_frames = 8, 4
    rec = torch.full((ncap * max_frames * sdist.FRAME_BYTES,), rank + 1, dtype=torch.uint8)
    cnt = torch.full((ncap,), 2, dtype=torch.int32)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        all_rec, all_cnt = sdist.gather_frames(rec, cnt, world)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(all_cnt.sum()) != 2 * ncap * world or not bool(torch.equal(all_rec[rank * rec.numel():(rank + 1) * rec.numel()], rec)):
        raise SystemExit("rank %d: rehearsal gather wrong" % rank)
    # the per-rank kernel means travel as in the real run (one all_gather of a float per rank); made up here
    allk = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allk, torch.tensor([0.5 + 0.01 * rank], dtype=torch.float64))
    rank_fk = [float(v.item()) for v in allk]
    dist.destroy_process_group()
    if rank == 0:
        line = {"metric": "IQ Msps through filter+demod+slice+decode (whole job)", "value": None, "unit": "Msps",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(float(t.item()) / max(args.steps, 1) * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "rehearsal": "gloo ranks on CPU, no chain work, nothing measured",
                "roofline": {"kernel_ms_per_rank": rank_fk, "frac_min": None, "frac_max": None}, "cpu_baseline": None}
        if not args.no_cpu_baseline:
            # the CPU leg as in the real run: rank 0 alone, after the process group is gone, on captures made on the host
            from sdr_amd import synth
            caps = np.stack([synth.ts33c_capture(N_PER_CAPTURE, FS, seed=33 + i, f_off=F_OFF)[0].view(np.float32).reshape(-1, 2)
                             for i in range(2)])
            taps = np.asarray(_oracle_low_pass(), np.float32)
            line["cpu_baseline"] = cpu_baseline(caps, taps)
        print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--captures", type=int, default=1250, help="captures per GPU (configs[3]: 10000 / 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", nargs=4, metavar=("NPY", "FIRST", "COUNT", "SECONDS"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker[0], int(args.cpu_worker[1]), int(args.cpu_worker[2]), float(args.cpu_worker[3]))
    rehearsal = os.environ.get("SDR_BENCH_REHEARSAL") == "gloo"
    # SDR_BENCH_REHEARSAL=gloo-gpu: the real multi-rank run (every workload, every kernel, every check) with N ranks
    # that SHARE GPU 0 and a gloo process group whose collectives are staged through the host -- what a one-GPU box
    # can show of the N > 1 path.  Its timings mean nothing and the line says so.
    shared_gpu = os.environ.get("SDR_BENCH_REHEARSAL") == "gloo-gpu"
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:], rehearsal or shared_gpu))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    if rehearsal:
        return rehearsal_main(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))

    import torch
    import torch.distributed as dist
    from sdr_amd import _lib, chain
    from sdr_amd import dist as sdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if shared_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # One real stream for everything this process queues: torch's default stream has handle 0, which
    # sdr_set_stream reads as "the chain's own stream" -- the chains would then run beside torch's copies, event
    # records and the RCCL collectives (ordered on torch's current stream) instead of in order with them.
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    # SDR_BENCH_DIST=1 under torchrun with ONE rank walks the multi-GPU code (RCCL init, frame gather,
    # barriers, max over ranks) on a one-GPU box
    use_dist = world > 1 or (os.environ.get("SDR_BENCH_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if use_dist and shared_gpu:
        dist.init_process_group("gloo")
        stage_collectives_through_host(torch, dist)
    elif use_dist:
        dist.init_process_group("nccl", device_id=dev)

    ncap = args.captures
    iq, pkts = make_batch(torch, dev, ncap, seed=1000 + rank)
    taps = chain.low_pass(1.0, FS, 20e3, 40e3)
    max_frames = 4
    c = chain.Chain(FS, decim=DECIM, center_freq=F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_batch=ncap,
                    max_frames=max_frames, max_samples=N_PER_CAPTURE)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    rec = torch.zeros(ncap * max_frames * 96, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(ncap, dtype=torch.int32, device=dev)
    if use_dist:
        # The frame gather of step k runs on its own stream while the front end of step k+1 runs on the
        # main one: frame records are exported (a small copy kernel on the stream results are ordered on) into one
        # of two buffer pairs, the RCCL all_gather of that pair is queued behind an event on the side stream,
        # and a pair is reused only after its gather has finished.
        comm = torch.cuda.Stream(device=dev)
        recs, cnts = [rec, torch.zeros_like(rec)], [cnt, torch.zeros_like(cnt)]
        all_recs = [torch.zeros(world * rec.numel(), dtype=torch.uint8, device=dev) for _ in range(2)]
        all_cnts = [torch.zeros(world * ncap, dtype=torch.int32, device=dev) for _ in range(2)]
        ev_exported = [torch.cuda.Event() for _ in range(2)]
        ev_gathered = [torch.cuda.Event() for _ in range(2)]
    kstep = [0]

    res_stream = torch.cuda.current_stream()   # the handle's stream: what sdr_export_frames writes is ordered on it
    # Multi-rank runs: back end, frame export and gather all leave the main stream (sdr_set_overlap: the export is
    # ordered on the handle's result stream), so the main stream holds front ends only -- one rank through RCCL,
    # 20 steps: 0.470 ms per step with the export on the main stream, 0.457 this way (0.448 without any gather).
    # SDR_BENCH_DIST_OVERLAP=0 keeps everything but the gather on one stream.
    dist_overlap = use_dist and os.environ.get("SDR_BENCH_DIST_OVERLAP", "1") != "0"
    if dist_overlap:
        c.set_overlap(True)
        res_stream = torch.cuda.ExternalStream(c.result_stream(), device=dev)

    def step():
        c.batch_process(iq)
        if use_dist:
            b = kstep[0] & 1
            kstep[0] += 1
            res_stream.wait_event(ev_gathered[b])
            c.export_frames(recs[b], cnts[b])
            ev_exported[b].record(res_stream)
            with torch.cuda.stream(comm):
                comm.wait_event(ev_exported[b])
                sdist.gather_frames(recs[b], cnts[b], world, all_recs[b], all_cnts[b])
                ev_gathered[b].record(comm)

    # untimed pre-warm (clocks): by wall time on one GPU; a fixed count when step() holds a collective, so that every rank
    # calls it equally often
    if use_dist:
        for _ in range(300 if PREWARM_S > 0 else 0):
            step()
        torch.cuda.synchronize()
    else:
        prewarm(torch, step)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    c.profile_enable(True, PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    front_ms, back_ms = c.profile_read()
    c.profile_enable(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if use_dist:
        # what the last step gathered: every rank's counts and records, this rank's own slice intact
        b = (kstep[0] - 1) & 1
        if int(all_cnts[b].sum().item()) != 2 * ncap * world:
            raise SystemExit("rank %d: gathered frame counts sum to %d, expected %d" % (rank, int(all_cnts[b].sum().item()), 2 * ncap * world))
        mine = all_recs[b][rank * rec.numel():(rank + 1) * rec.numel()]
        if not bool(torch.equal(mine, recs[b])) or not bool(torch.equal(all_cnts[b][rank * ncap:(rank + 1) * ncap], cnts[b])):
            raise SystemExit("rank %d: own slice of the gathered frame table differs" % rank)
    # ---- correctness of what was timed (size-independent property + oracle spot check)
    frames = c.fetch_frames()
    got = {}
    for f in frames:
        if f.status == _lib.FRAME_OK:
            got.setdefault(f.capture, []).append(f.packet())
    n_ok = sum(1 for i in range(ncap) if got.get(i) == [pkts[i], pkts[i]])
    if n_ok != ncap:
        raise SystemExit("rank %d: only %d of %d captures decoded to the transmitted packets" % (rank, n_ok, ncap))
    if rank == 0:
        from oracle import decoders as odec
        from oracle import dsp as odsp
        bits = c.fetch_bits().reshape(ncap, -1)
        for i in (0, ncap // 2, ncap - 1):
            x = iq[i].cpu().numpy().view(np.complex64).reshape(-1)
            want = odsp.ook_chain(x, taps, DECIM, 1.0, 0.05, 0.15, 1, fc=F_OFF, fs=FS)
            if not np.array_equal(bits[i], want):
                bad = np.flatnonzero(bits[i] != want)
                raise SystemExit("capture %d: sliced bits differ from the oracle at %d of %d positions (first %s; "
                                 "ones: got %d, oracle %d)"
                                 % (i, bad.size, want.size, bad[:8].tolist(), int(bits[i].sum()), int(want.sum())))
            d = odec.TS33CStreamDecoder(sample_rate=20000)
            d.work([want])
            if [f["packet"] for f in d.frames] != got.get(i):
                raise SystemExit("capture %d: frames differ from the oracle" % i)

    samples_per_step = ncap * N_PER_CAPTURE
    total_samples = samples_per_step * args.steps * world
    value = total_samples / elapsed / 1e6
    fk = float(np.mean(front_ms)) if front_ms else float("nan")
    bk = float(np.mean(back_ms)) if back_ms else float("nan")
    # every rank's mean kernel time (one all_gather of a float per rank): the roofline of the slowest and the
    # fastest GPU stand next to rank 0's
    rank_fk = [fk]
    if use_dist:
        mine = torch.tensor([fk], dtype=torch.float64, device=dev)
        allk = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allk, mine)
        rank_fk = [float(v) for v in allk.cpu().numpy()]
    cpu_caps = cpu_sample(iq) if (rank == 0 and not args.no_cpu_baseline) else None
    line = None
    if rank == 0:
        achieved = BYTES_PER_SAMPLE * samples_per_step / (fk * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic("k_frontend<true, false> fc32", "captures_per_gpu=%d" % ncap)
        line = {
            "metric": "IQ Msps through filter+demod+slice+decode (whole job)",
            "value": round(value, 1), "unit": "Msps", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "prewarm": ("%g s of the same step, untimed, before the warm-up steps (clock settling)" % PREWARM_S) if not use_dist
                       else ("300 untimed steps before the warm-up steps (clock settling)" if PREWARM_S > 0 else "off"),
            "data": "synthetic (TS33C OOK/PWM frames, random payloads, carrier +100 kHz, AWGN; generated on the GPU)",
            "config": {"workload": "ook_batch_ts33c_xlating", "captures_per_gpu": ncap,
                       "samples_per_capture": N_PER_CAPTURE, "sample_rate": FS, "ntaps": int(taps.size),
                       "decim": DECIM, "center_freq": F_OFF, "chain": "xlating_fir->mag2->threshold->slicer->ts33c_decoder",
                       "parallelism": "captures sharded per GPU, RCCL all_gather of frame records" if world > 1 else "1 GPU",
                       "streams": ("front ends on the main stream; back end + frame export on the handle's result stream "
                                   "(sdr_set_overlap); all_gather on a third") if dist_overlap else "one stream"},
            "roofline": {"bound": "hbm", "kernel": "k_frontend<true, false>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": round(fk, 4), "backend_kernel_ms": round(bk, 4),
                         "bytes_per_launch": BYTES_PER_SAMPLE * samples_per_step,
                         "kernel_timing": "HIP events on the handle's stream around every %d%s step of the timed region" % (PROFILE_EVERY, "rd" if PROFILE_EVERY == 3 else "th")},
            "frames_decoded": len(frames) * world, "verified": "all captures decode to the transmitted packets; 3 captures bit-exact vs oracle",
        }
        if shared_gpu:
            line["rehearsal"] = "%d gloo ranks sharing GPU 0, collectives staged through the host: a check of the N > 1 path, no timing here is a measurement" % world
        per_rank = [BYTES_PER_SAMPLE * samples_per_step / (k * 1e-3) / 1e9 / HBM_PEAK_GBS for k in rank_fk]
        line["roofline"].update({"frac_min": round(min(per_rank), 4), "frac_max": round(max(per_rank), 4),
                                 "kernel_ms_per_rank": [round(k, 4) for k in rank_fk]})
        line["cpu_baseline"] = None
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cpu_caps, taps)
        if world == 1 and not use_dist:
            progress("ook_batch_cu8")
            other = {"ook_batch_cu8": cu8_workload(torch, dev, iq, pkts, taps)}
            del iq
            torch.cuda.empty_cache()
            progress("fm_radio_wbfm")
            other["fm_radio_wbfm"] = fm_workload(torch, dev, args.no_cpu_baseline)
            progress("ook_r_shape_ts33c")
            other["ook_r_shape_ts33c"] = r_shape_workload(torch, dev, args.no_cpu_baseline)
            progress("wideband_pfb64")
            other["wideband_pfb64"] = pfb_workload(torch, dev, args.no_cpu_baseline)
            progress("tfa_fsk_discriminator")
            other["tfa_fsk_discriminator"] = tfa_fsk_workload(torch, dev, args.no_cpu_baseline)
            progress("ook_batch_somfy_berner_mumbi")
            other["ook_batch_somfy_berner_mumbi"] = mixed_emitters_workload(torch, dev, args.no_cpu_baseline)
            line["other_workloads"] = other
    c.close()
    if use_dist:
        # every rank: one wideband capture cut into `world` segments (configs[4] on N GPUs)
        del iq
        torch.cuda.empty_cache()
        if rank == 0:
            progress("wideband_pfb64_segments")
        seg = pfb_segments_workload(torch, dist, dev, rank, world)
        if rank == 0:
            progress("long_capture_frames_only")
        lc = long_capture_frames_workload(torch, dist, dev, rank, world)
        if rank == 0:
            line["other_workloads"] = {"wideband_pfb64_segments": seg, "long_capture_frames_only": lc}
        dist.destroy_process_group()
        # the CPU leg runs on rank 0 alone, after the process group is gone: no rank waits inside a collective for it
        if rank == 0 and not args.no_cpu_baseline and world > 1:
            line["cpu_baseline"] = cpu_baseline(cpu_caps, taps)
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
