set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_conv_spec_gpu.py tests/test_conv_f16x3_gpu.py tests/test_conv_sweep_gpu.py tests/test_fullsize_reference_gpu.py tests/test_adn_fused_gpu.py -x -q -m gpu > $O/ab9_tests.log 2>&1 || (tail -60 $O/ab9_tests.log | cut -c1-300; exit 1)
tail -2 $O/ab9_tests.log
timeout -k 10 900 python tools/ab_lib.py adell_mri_amd/libadellhip_base.so adell_mri_amd/libadellhip.so 3 16 > $O/ab9_step.log 2>&1
cat $O/ab9_step.log
