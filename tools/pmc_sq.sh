#!/bin/bash
# SQ counter passes (one rocprofv3 run each) over tools/exp_front.py (or tools/$2); usage: SDR_D=5 tools/pmc_sq.sh <tag> [script.py [kernel substring]]
# writes gpurun_out/<tag>_sq.csv: counter, mean per launch of the matching kernel
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
prog=${2:-exp_front.py}
export KSUB=${3:-k_frontend}
out=$R/gpurun_out/${tag}_sq.csv
echo "counter,mean_per_launch,launches" > $out
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  d=/tmp/pmc_$$_$RANDOM
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
