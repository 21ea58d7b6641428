set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PRE=fm timeout -k 10 60 python3 $R/tools/exp_ov_hang.py torch 1 3 2>&1 | grep -v amdgpu.ids | tail -4
cd $R
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/r03_f_gpu_tests.log 2>&1 || { tail -30 $O/r03_f_gpu_tests.log; exit 1; }
tail -2 $O/r03_f_gpu_tests.log
cd /tmp
echo "== overlap A/B"
timeout -k 10 200 python3 $R/tools/exp_overlap_ab.py 2>&1 | grep overlap
echo "== profile round"
bash $R/tools/profile_round.sh r03_b
