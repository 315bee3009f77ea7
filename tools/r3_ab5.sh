set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_adn_fused_gpu.py tests/test_conv_spec_gpu.py tests/test_conv_f16x3_gpu.py -x -q -m gpu > $O/ab5_tests.log 2>&1 || (tail -60 $O/ab5_tests.log | cut -c1-300; exit 1)
tail -2 $O/ab5_tests.log
timeout -k 10 600 python tools/ab_step.py hf:no_adn_fuse 1 5 8 > $O/ab5_adn_step.log 2>&1
cat $O/ab5_adn_step.log
timeout -k 10 600 python tools/ab_step.py igemm_wide8 1 5 8 > $O/ab5_wide8_step.log 2>&1
cat $O/ab5_wide8_step.log
ADELL_IGEMM_WIDE8=1 timeout -k 10 600 python -m pytest tests/test_conv_spec_gpu.py tests/test_fullsize_reference_gpu.py -x -q -m gpu -k "spec or c_oracle" > $O/ab5_wide8_tests.log 2>&1 || (tail -40 $O/ab5_wide8_tests.log | cut -c1-300; exit 1)
tail -2 $O/ab5_wide8_tests.log
