# HBM traffic per launch of the bench workload's kernels (the two --pmc passes of profile_round.sh alone):
# bash tools/prof_traffic.sh <tag>
set -e
TAG=${1:-traffic}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
export ADELL_WGRAD_STREAM=0
A="--steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $A > $O/bench_line.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -- python3 $R/bench.py $A > $O/pmcF.json 2> $O/pmcF.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -- python3 $R/bench.py $A > $O/pmcW.json 2> $O/pmcW.err
cd $R
python3 tools/pmc_traffic.py $O/pmcF $O/pmcW $O/pmc_traffic.json $O/bench_line.json > /dev/null 2>&1 || true
find $O -name "*.csv" -size +3M -delete
python3 - <<PY
import json
d = json.load(open("$O/pmc_traffic.json"))
for k in ("adell_conv_wgrad_zring_kernel", "adell_conv_igemm_f16_kernel", "adell_fwd_s2_fused_kernel"):
    v = d["kernels"].get(k)
    print(k, v and (v["launches"], round(v["traffic_bytes_mean"] / 1e6, 1)))
PY
