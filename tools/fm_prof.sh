#!/bin/bash
# FM chain: GPU tests, then a per-kernel rocprofv3 breakdown of tools/exp_fm.py -> gpurun_out/<tag>_fm_by_grid.csv
tag=${1:-fm}
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 240 python -u -m pytest tests/test_gpu_fm.py tests/test_gpu_mfma.py -x -q --timeout 120 --timeout-method thread > gpurun_out/${tag}_t.log 2>&1
rc=$?
tail -3 gpurun_out/${tag}_t.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fmprof
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fmprof -o fm -- python3 $R/tools/exp_fm.py > $R/gpurun_out/${tag}_e.log 2>&1 || exit 1
python3 $R/tools/trace_by_grid.py $(find /tmp/fmprof -name "*kernel_trace.csv") $R/gpurun_out/${tag}_fm_by_grid.csv
python3 - $R/gpurun_out/${tag}_fm_by_grid.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-28s grid %9s calls %3s avg %9.1f us" % (r["Name"][:28], r["GridX"], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep "FM 2" $R/gpurun_out/${tag}_e.log
