#!/bin/bash
# MFMA pipe counters of one kernel: usage tools/pmc_mfma.sh <tag> <script.py> <kernel substring>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; prog=$2; export KSUB=$3
out=$R/gpurun_out/${tag}_mfma.csv
echo "counter,mean_per_launch,launches" > $out
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LEVEL_WAVES"; do
  d=/tmp/pmcm_$$_$RANDOM
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/tools/$prog > /dev/null 2>&1
  python3 - $d $out <<'PY'
import csv, glob, sys, collections, os
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["KSUB"] in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "a") as o:
    for k in sorted(d):
        o.write("%s,%d,%d\n" % (k, sum(d[k]) / len(d[k]), len(d[k])))
PY
done
cat $out
