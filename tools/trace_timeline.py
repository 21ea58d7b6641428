"""rocprofv3 --kernel-trace CSV -> the last N launches in start order with the idle gap before each
(where a step's time goes between its kernels).  usage: trace_timeline.py kernel_trace.csv [N]"""
import csv, sys


def main(src, n=40):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40],
             r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Queue_Id", "")) for r in csv.DictReader(open(src))]
    rows.sort()
    rows = rows[-int(n):]
    prev = rows[0][0]
    t0 = rows[0][0]
    for s, e, name, grid, q in rows:
        print("start %9.1f  %9.1f us gap  %9.1f us  %-40s grid %-8s queue %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name, grid, q))
        prev = max(prev, e)


if __name__ == "__main__":
    main(*sys.argv[1:])
