set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/r03_c_gpu_tests.log 2>&1 || { tail -30 $O/r03_c_gpu_tests.log; exit 1; }
tail -2 $O/r03_c_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
echo "== one-rank RCCL walk of the multi-GPU bench"
SDR_BENCH_DIST=1 SDR_BENCH_CPU_BUDGET=2 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 1 --steps 10 --warmup 2 > $O/r03_c_bench_dist1.json 2> $O/r03_c_bench_dist1.err || { tail -20 $O/r03_c_bench_dist1.err; exit 1; }
cut -c1-600 $O/r03_c_bench_dist1.json
echo "== profile round"
bash $R/tools/profile_round.sh r03_a
