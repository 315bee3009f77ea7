set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_adn_fused_gpu.py -x -q -m gpu > $O/ab2_adn_tests.log 2>&1 || (tail -40 $O/ab2_adn_tests.log; exit 1)
tail -3 $O/ab2_adn_tests.log
timeout -k 10 600 python tools/ab_step.py hf:no_adn_fuse 1 5 8 > $O/ab2_adn_step.log 2>&1
cat $O/ab2_adn_step.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/ab2_all_tests.log 2>&1 || (tail -40 $O/ab2_all_tests.log; exit 1)
tail -3 $O/ab2_all_tests.log
