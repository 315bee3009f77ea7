set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_conv_cinfold_gpu.py tests/test_unet_gpu.py tests/test_tokens_gpu.py -x -q -m gpu > $O/ab7_tests.log 2>&1 || (tail -60 $O/ab7_tests.log | cut -c1-300; exit 1)
tail -2 $O/ab7_tests.log
timeout -k 10 300 python tools/small_trio_time.py 128 2 2>/dev/null | tail -1
