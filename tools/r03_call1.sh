# round 3, first GPU call: suite, baseline bench, then the --pmc stall diagnostics (each step under its own limit, chained)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/r03_a_gpu_tests.log 2>&1 || { tail -30 $O/r03_a_gpu_tests.log; exit 1; }
tail -2 $O/r03_a_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/r03_a_bench.json 2> $O/r03_a_bench.err
echo "bench ok"
P=$R/tools/probe/two_queue_pmc
timeout -s KILL 60 $P 40 own 1 4 > $O/r03_a_two_queue.log 2>&1
for v in "own 1 4" "null 1 4" "null 0 4" "null 1 1"; do
  echo "== rocprofv3 --pmc FETCH_SIZE -- two_queue_pmc 40 $v" >> $O/r03_a_two_queue.log
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tq_$RANDOM -- $P 40 $v >> $O/r03_a_two_queue.log 2>&1 || { echo "exit $? for $v" >> $O/r03_a_two_queue.log; tail -5 $O/r03_a_two_queue.log; exit 1; }
done
tail -12 $O/r03_a_two_queue.log
echo "== stall script under --pmc"
SDR_OV_TRACE=1 timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_stall -- python3 $R/tools/exp_pmc_stall.py > $O/r03_a_pmc_stall.log 2>&1 || { echo "stall script exit $?"; tail -40 $O/r03_a_pmc_stall.log; exit 1; }
tail -5 $O/r03_a_pmc_stall.log | cut -c1-300
