# EXPERIMENT: which Linear-layer GEMMs go to the f16x3 kernel (ops.FLAGS gemm_f16x3 + size knobs):
# VICReg ConvNeXt / UNETR / SWIN-UNet step times per rule, alternating in one call.
run() { echo "== $1"; shift; env "$@" python tools/bench_ssl.py --batch 32 --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ssl', round(d['ms_per_step'],2))"; env "$@" python tools/bench_unetr.py --steps 20 --warmup 4 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unetr', round(d['ms_per_step'],2))"; env "$@" python tools/bench_swin.py --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('swin', round(d['ms_per_step'],2))"; }
run off X=1
run mn96 ADELL_GEMM_F16X3=1 ADELL_GEMM_F16X3_MIN_MN=96
run mn64 ADELL_GEMM_F16X3=1 ADELL_GEMM_F16X3_MIN_MN=64
run k192mn96 ADELL_GEMM_F16X3=1 ADELL_GEMM_F16X3_MIN_K=192 ADELL_GEMM_F16X3_MIN_MN=96
run off2 X=1
run mn96b ADELL_GEMM_F16X3=1 ADELL_GEMM_F16X3_MIN_MN=96
