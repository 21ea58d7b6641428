# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# bench line, rocprofv3 kernel stats (plain and per grid) of the same command, FETCH_SIZE / WRITE_SIZE passes
set -e
tag=${1:-r01_x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_rocprof.err
cp $(ls -S $(find /tmp/prof_$tag -name "*kernel_stats.csv") | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
python3 $R/tools/trace_by_grid.py $(ls -S $(find /tmp/prof_$tag -name "*kernel_trace.csv") | head -1) $R/gpurun_out/${tag}_kernel_stats_by_grid.csv
# counter passes (the overlapped second passes run here too since round 3: DESIGN 4e), output kept
export SDR_BENCH_PREWARM=0   # counters are per launch: no need for settled clocks, and serialised launches are slow
echo "FETCH_SIZE pass" && timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1
echo "WRITE_SIZE pass" && timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_write.log 2>&1
cp $(ls -S $(find /tmp/pmc_f_$tag -name "*counter_collection.csv") | head -1) $R/gpurun_out/${tag}_pmc_fetch.csv
cp $(ls -S $(find /tmp/pmc_w_$tag -name "*counter_collection.csv") | head -1) $R/gpurun_out/${tag}_pmc_write.csv
python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}_pmc_fetch.csv $R/gpurun_out/${tag}_pmc_write.csv $R/gpurun_out/${tag}_pmc.json
head -12 $R/gpurun_out/${tag}_kernel_stats_by_grid.csv
