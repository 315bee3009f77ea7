set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_augment.py tests/test_unet_gpu.py tests/test_inference.py tests/test_zz_multirank_gpu.py tests/test_adn_fused_gpu.py tests/test_handoff.py tests/test_skip_fork_gpu.py -x -q -m gpu > $O/ab4_tests.log 2>&1 || (tail -60 $O/ab4_tests.log | cut -c1-300; exit 1)
tail -3 $O/ab4_tests.log
timeout -k 10 600 python tools/ab_step.py hf:no_adn_fuse 1 5 8 > $O/ab4_adn_step.log 2>&1
cat $O/ab4_adn_step.log
