set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_wgrad_s2_gpu.py tests/test_dgrad_s2_fused_gpu.py tests/test_fwd_s2_fused_gpu.py tests/test_conv_sweep_gpu.py tests/test_conv_f16x3_gpu.py tests/test_conv_spec_gpu.py tests/test_skip_fork_gpu.py tests/test_convt_k2_gpu.py -x -q -m gpu > $O/ab1_tests.log 2>&1
tail -3 $O/ab1_tests.log
for lib in libadellhip_base.so libadellhip.so libadellhip_rv.so; do
  ADELL_HIP_LIBRARY=$R/adell_mri_amd/$lib timeout -k 10 300 python tools/small_trio_time.py 128 2 >> $O/ab1_trio.log 2>&1
done
cat $O/ab1_trio.log
timeout -k 10 600 python tools/ab_lib.py adell_mri_amd/libadellhip_base.so adell_mri_amd/libadellhip.so 3 16 > $O/ab1_step.log 2>&1
cat $O/ab1_step.log
