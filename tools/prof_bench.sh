# kernel-trace summary of the headline step (one stream): bash tools/prof_bench.sh <tag>
set -e
TAG=${1:-bench}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
export ADELL_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary > $O/bench_line.json 2> $O/stats.err || true
cd $R
python3 tools/trace_stats.py $O/stats > $O/kernel_stats.txt 2>&1 || true
find $O -name "*.csv" -size +3M -delete
