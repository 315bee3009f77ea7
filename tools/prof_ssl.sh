# kernel-trace summary of the config-4 step (one stream): bash tools/prof_ssl.sh <tag> [extra env assignments...]
set -e
TAG=${1:-ssl}
shift || true
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
export ADELL_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ssl -- python3 $R/tools/bench_ssl.py --batch 32 --steps 6 --warmup 2 > $O/ssl_line.json 2> $O/ssl.err || true
cd $R
python3 tools/trace_stats.py $O/ssl > $O/ssl_convnext_kernel_stats.txt 2>&1 || true
find $O -name "*.csv" -size +3M -delete
head -45 $O/ssl_convnext_kernel_stats.txt
