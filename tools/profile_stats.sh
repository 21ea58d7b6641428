set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -- python3 $R/bench.py > $R/gpurun_out/r01_m_bench_under_rocprof.json 2> $R/gpurun_out/r01_m_rocprof.err
cp $(find /tmp/prof_m -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_m_kernel_stats.csv
python3 $R/tools/trace_by_grid.py $(find /tmp/prof_m -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r01_m_kernel_stats_by_grid.csv
head -1 $(find /tmp/prof_m -name "*kernel_trace.csv" | head -1)
head -14 $R/gpurun_out/r01_m_kernel_stats_by_grid.csv
