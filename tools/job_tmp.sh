set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 500 python -m pytest tests/test_adn_fused_gpu.py tests/test_unet_gpu.py tests/test_ops_gpu.py tests/test_layers_sweep_gpu.py -x -q -m gpu > gpurun_out/r3/lr_tests.log 2>&1 || (tail -60 gpurun_out/r3/lr_tests.log | cut -c1-300; exit 1)
tail -3 gpurun_out/r3/lr_tests.log
timeout -k 10 400 python tools/ab_step.py hf:no_lowrank 1 6 8 2>&1 | tail -1
