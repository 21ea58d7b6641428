"""rocprofv3 --kernel-trace CSV -> per (kernel, grid, workgroup) launch statistics.
The same kernel instantiation serves several bench workloads (k_frontend<true, false> runs the
headline batch at decimation 50 and the mixed-emitter batch at decimation 10): the plain --stats
summary averages them together, this one keeps them apart.  usage: trace_by_grid.py kernel_trace.csv out.csv"""
import collections, csv, statistics, sys


def main(src, dst):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(src)):
        key = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""),
               r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridX", "GridY", "WorkgroupX", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for (name, gx, gy, wg), v in rows:
            if name.startswith("k_") or "k_frontend" in name or "k_pfb" in name or "k_fir" in name or "k_fm" in name:
                w.writerow([name, gx, gy, wg, len(v), sum(v), round(statistics.mean(v), 1), min(v), max(v)])


if __name__ == "__main__":
    main(*sys.argv[1:3])
