#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS per wave of k_frontend_win for a few shapes (tools/exp_win.py via WIN_ARGS); one rocprofv3 run each
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-r03_win_variants}.csv
echo "variant,counter,mean_per_launch,launches" > $out
for v in "27 quad 1 64 20 win" "27 mag2 1 64 20 win" "1 mag2 1 64 20 win" "1 quad 1 64 20 win" "121 mag2 100 64 19 win"; do
  export WIN_ARGS="$v"
  d=/tmp/pmcw_$$_$RANDOM
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $d -- python3 $R/tools/exp_win.py > /dev/null 2>&1
  python3 - $d $out "$v" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_frontend_win" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "a") as o:
    for k in sorted(d):
        o.write("%s,%s,%d,%d\n" % (sys.argv[3], k, sum(d[k]) / len(d[k]), len(d[k])))
PY
done
cat $out
