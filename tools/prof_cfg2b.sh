# kernel-trace summary of the config-2b step (ResNet-backbone U-Net, one stream): bash tools/prof_cfg2b.sh <tag>
set -e
TAG=${1:-cfg2b}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
export ADELL_WGRAD_STREAM=0
export ADELL_BENCH_ONLY=cfg2b_resnet_backbone_128
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg2b -- python3 $R/tools/secondary_only.py > $O/cfg2b_line.txt 2> $O/cfg2b.err || true
cd $R
python3 tools/trace_stats.py $O/cfg2b > $O/backbone_unet_kernel_stats.txt 2>&1 || true
find $O -name "*.csv" -size +3M -delete
head -50 $O/backbone_unet_kernel_stats.txt
