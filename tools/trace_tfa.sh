R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SDR_BENCH_PREWARM=0.02 timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/tfatr -- python3 $R/tools/exp_tfa.py 2>&1 | grep Msps
python3 $R/tools/trace_timeline.py $(ls -S $(find /tmp/tfatr -name "*kernel_trace.csv") | head -1) 60
