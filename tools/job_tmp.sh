set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
true
true
timeout -k 10 300 python tools/ab_step.py igemm_bglobal 1 6 8 2>&1 | tail -1
