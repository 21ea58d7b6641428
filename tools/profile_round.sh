set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/r01_m_bench.json 2> $R/gpurun_out/r01_m_bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -- python3 $R/bench.py > $R/gpurun_out/r01_m_bench_under_rocprof.json 2> $R/gpurun_out/r01_m_rocprof.err
f=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r01_m_kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $R/gpurun_out/r01_m_pmc_fetch.csv
cp $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $R/gpurun_out/r01_m_pmc_write.csv
head -12 $R/gpurun_out/r01_m_kernel_stats.csv
