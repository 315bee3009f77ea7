# kernel-trace summary of the config-5 step (one stream): bash tools/prof_swin.sh <tag>
set -e
TAG=${1:-swin}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
export ADELL_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/swin -- python3 $R/tools/bench_swin.py --steps 4 --warmup 2 > $O/swin_line.json 2> $O/swin.err || true
cd $R
python3 tools/trace_stats.py $O/swin > $O/swinunet_kernel_stats.txt 2>&1 || true
find $O -name "*.csv" -size +3M -delete
head -40 $O/swinunet_kernel_stats.txt
