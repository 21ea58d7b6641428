"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv (two passes over
`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`) -> profiles/<tag>_pmc.json
with per-launch HBM bytes per kernel (gfx950 correction: FETCH_SIZE x 2 for 16-B/lane
streaming reads, MI355X_MICROARCH.md HBM section).

One kernel instantiation serves several bench workloads (k_frontend_win: TFA batch, R shape, FM stage B), so a
workload's launches are the (kernel, grid, workgroup) group whose corrected fetch is closest to the workload's
algorithmic read bytes.  `csrc_sha256` is the hash of the kernel sources the passes ran on: bench.py compares it
with the tree it runs from and says so next to `traffic` (counters cannot be read from inside the process)."""
import collections, csv, hashlib, json, math, os, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sdr_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"])].append(float(r["Counter_Value"]))
    return d


def main(fetch_csv, write_csv, out_json):
    f, w = agg(fetch_csv), agg(write_csv)
    n = 1250 * (1 << 18)
    fm_n = 1 << 26
    # (kernel substring, key, algorithmic bytes, algorithmic READ bytes that pick the launch group, workload,
    #  fetch correction: 2 = 16-B/lane streaming reads (calibrated), 1 = other widths (uncalibrated, reported as counted))
    want = [("k_frontend<true, false,", "k_frontend<true, false> fc32", 8.02 * n, 8.0 * n, "ook_batch_ts33c_xlating captures_per_gpu=1250", 2),
            ("k_frontend<true, false,", "k_frontend<true, false> fc32 D=10", 8.1 * n, 8.0 * n * 1.024, "ook_batch_somfy_berner_mumbi captures=1250", 2),
            ("k_frontend_raw8", "k_frontend_raw8 cu8", 2.02 * n, 2.0 * n, "ook_batch_cu8 captures_per_gpu=1250", 2),
            ("k_pfb<8>", "k_pfb<8>", (8 + 1 / 16) * (1 << 26), 8.0 * (1 << 26), "wideband_pfb64 samples=2^26", 2),
            ("k_frontend_win", "k_frontend_win tfa", 9.0 * 64 * (1 << 20), 8.0 * 64 * (1 << 20), "tfa_fsk_discriminator captures=64", 2),
            ("k_frontend_win", "k_frontend_win r_shape", 8.01 * 64 * (1 << 19), 8.0 * 64 * (1 << 19), "ook_r_shape_ts33c captures=64", 2),
            # FM chain, 2^26 input samples: stage A reads them, writes 2^24 complex; stage B reads those, writes 2^24 floats; ...
            ("k_frontend_mfma2", "fm stage A k_frontend_mfma2", 8.0 * fm_n + 8.0 * fm_n / 4, 8.0 * fm_n, "fm_radio_wbfm samples=2^26", 2),
            ("k_frontend_win", "fm stage B k_frontend_win", 8.0 * fm_n / 4 + 4.0 * fm_n / 4, 8.0 * fm_n / 4, "fm_radio_wbfm samples=2^26", 2),
            ("k_fm_stage_ab", "fm stages A+B k_fm_stage_ab", 8.0 * fm_n + 4.0 * fm_n / 4, 8.0 * fm_n, "fm_radio_wbfm samples=2^26", 2),
            ("k_fir_fff_mfma", "fm audio FIR k_fir_fff_mfma", 4.0 * fm_n / 4 + 4.0 * fm_n / 40, 4.0 * fm_n / 4, "fm_radio_wbfm samples=2^26", 2),
            ("k_fm_audio", "fm audio FIR + de-emphasis + resampler k_fm_audio", 4.0 * fm_n / 4 + 4.0 * fm_n / 40 * 0.96, 4.0 * fm_n / 4, "fm_radio_wbfm samples=2^26", 2),
            ("k_deemph_resample", "fm de-emphasis + resampler k_deemph_resample", 4.0 * fm_n / 40 * 2.96, 4.0 * fm_n / 40, "fm_radio_wbfm samples=2^26", 1),
            ("k_deemph_warm", "fm de-emphasis k_deemph_warm", 8.0 * fm_n / 40, 4.0 * fm_n / 40, "fm_radio_wbfm samples=2^26", 1),
            ("k_resample_fff_lds", "fm resampler k_resample_fff_lds", 4.0 * fm_n / 40 * 1.96, 4.0 * fm_n / 40, "fm_radio_wbfm samples=2^26", 1),
            ("k_tail_update_batch", "fm history update k_tail_update_batch", 0.0, 1.0, "fm_radio_wbfm samples=2^26", 1)]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (separate passes), python3 bench.py "
                     "--steps 3 --warmup 1 --no-cpu-baseline; per-launch means",
           "correction": "gfx950: FETCH_SIZE counts the 128-B requests of 16-B/lane streaming reads at 64 B "
                         "(MI355X_MICROARCH.md, HBM) -> x2 on kernels that read that way; other widths are uncalibrated "
                         "and reported as counted (fetch_correction 1); WRITE_SIZE exact; KB = 1024 B",
           "csrc_sha256": csrc_sha256(),
           "kernels": {}}
    for sub, key, alg, alg_read, workload, corr in want:
        ks = [k for k in f if sub in k[0] and k in w]
        if not ks:
            continue
        if key.startswith("fm history"):
            k = max(ks, key=lambda k: len(f[k]))
        else:
            k = min(ks, key=lambda k: abs(math.log(max(statistics.mean(f[k]) * 1024 * corr, 1.0) / alg_read)))
        fb = statistics.mean(f[k]) * 1024 * corr
        wb = statistics.mean(w[k]) * 1024
        out["kernels"][key] = {"workload": workload, "kernel": k[0].split("(")[0], "grid_size": int(k[1]), "workgroup_size": int(k[2]),
                               "launches": len(f[k]), "FETCH_SIZE_KB": statistics.mean(f[k]),
                               "WRITE_SIZE_KB": statistics.mean(w[k]), "fetch_correction": corr, "fetch_bytes_corrected": fb, "write_bytes": wb,
                               "hbm_bytes": fb + wb, "algorithmic_bytes": alg,
                               "traffic_over_algorithmic": (fb + wb) / alg if alg else None}
    fm = [v for k, v in out["kernels"].items() if k.startswith("fm ")]
    if fm:
        alg = 8.08 * fm_n
        tot = sum(v["hbm_bytes"] for v in fm)
        out["kernels"]["fm chain"] = {"workload": "fm_radio_wbfm samples=2^26", "kernels": [v["kernel"] for v in fm],
                                      "hbm_bytes": tot, "algorithmic_bytes": alg, "traffic_over_algorithmic": tot / alg,
                                      "note": "sum over the chain's kernels; algorithmic = 8 B in + 4 B x 48e3/fs out per input sample (SURVEY 8d): "
                                              "the stage outputs written and re-read between the kernels are what exceeds it"}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, int(v["hbm_bytes"]), round(v["traffic_over_algorithmic"], 4) if v["traffic_over_algorithmic"] else "-")


if __name__ == "__main__":
    main(*sys.argv[1:4])
