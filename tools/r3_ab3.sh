set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/ab3_all_tests.log 2>&1 || (tail -60 $O/ab3_all_tests.log; exit 1)
tail -3 $O/ab3_all_tests.log
timeout -k 10 600 python tools/ab_step.py hf:no_adn_fuse 1 5 8 > $O/ab3_adn_step.log 2>&1
cat $O/ab3_adn_step.log
