set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_adn_fused_gpu.py tests/test_tokens_gpu.py tests/test_wgrad_zring_gpu.py tests/test_ops_gpu.py tests/test_conv_sweep_gpu.py -x -q -m gpu > $O/ab6_tests.log 2>&1 || (tail -60 $O/ab6_tests.log | cut -c1-300; exit 1)
tail -2 $O/ab6_tests.log
timeout -k 10 600 python tools/ab_step.py hf:no_adn_fuse 1 6 8 > $O/ab6_adn_step.log 2>&1
cat $O/ab6_adn_step.log
