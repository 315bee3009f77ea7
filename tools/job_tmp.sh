set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_layers_sweep_gpu.py tests/test_unet_gpu.py tests/test_adn_fused_gpu.py -x -q -m gpu > gpurun_out/r3/st_tests.log 2>&1 || (tail -60 gpurun_out/r3/st_tests.log | cut -c1-300; exit 1)
tail -2 gpurun_out/r3/st_tests.log
for l in base "" base "" base ""; do ADELL_HIP_LIBRARY=$PWD/adell_mri_amd/libadellhip${l:+_$l}.so timeout -k 10 200 python tools/ab_lib.py --child 24 | tail -1; done
