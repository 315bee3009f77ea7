# Round profile of the bench workload (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r03a').
# Four rocprofv3 runs of the SAME command, each with the program directly after `--`:
#   1. --kernel-trace --stats        per-kernel time
#   2. --pmc FETCH_SIZE              HBM read bytes   (separate passes: the TCC block cannot hold both)
#   3. --pmc WRITE_SIZE              HBM write bytes
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE   matrix-pipe busy + clock
# then the plain (un-profiled) default bench line. Summaries: tools/rocpd_stats.py,
# tools/pmc_traffic.py, tools/pmc_mfma.py -> copy into profiles/<tag>_*.
set -e
TAG=${1:-r03a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
ARGS="--steps 7 --warmup 3 --no-cpu-baseline --no-fp32 --no-secondary"
# (per-kernel passes with every launch on ONE stream -- ADELL_WGRAD_STREAM=0, exported, not `env`
# after `--` -- so that a kernel's duration and counters are its own; bench.py's event-timed steps
# do the same. Pass 1b is the default two-stream step, for the record of what overlaps.)
export ADELL_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/bench_line.json 2> $O/stats.err
export ADELL_WGRAD_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 $R/bench.py $ARGS > $O/bench_line_two_streams.json 2> $O/stats2.err
export ADELL_WGRAD_STREAM=0
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-secondary > $O/pmcF.json 2> $O/pmcF.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-secondary > $O/pmcW.json 2> $O/pmcW.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcM -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-secondary > $O/pmcM.json 2> $O/pmcM.err
cd $R
unset ADELL_WGRAD_STREAM
python3 tools/trace_stats.py $O/stats > $O/kernel_stats.txt 2>&1 || true
python3 tools/trace_stats.py $O/stats2 > $O/kernel_stats_two_streams.txt 2>&1 || true
python3 tools/pmc_traffic.py $O/pmcF $O/pmcW $O/pmc_traffic.json $O/bench_line.json > /dev/null 2>&1 || true
python3 tools/pmc_mfma.py $O/pmcM > $O/pmc_mfma_utilisation.txt 2>&1 || true
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err
# the driver's exact command as well (20 timed steps)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
# BASELINE configs 3 / 4 / 5: kernel summaries of their own step (one stream), program directly after `--`
cd /tmp
export ADELL_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unetr -- python3 $R/tools/bench_unetr.py --steps 6 --warmup 2 > $O/unetr_line.json 2> $O/unetr.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ssl -- python3 $R/tools/bench_ssl.py --batch 32 --steps 6 --warmup 2 > $O/ssl_line.json 2> $O/ssl.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/swin -- python3 $R/tools/bench_swin.py --steps 4 --warmup 2 > $O/swin_line.json 2> $O/swin.err || true
export ADELL_BENCH_ONLY=cfg2b_resnet_backbone_128
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg2b -- python3 $R/tools/secondary_only.py > $O/cfg2b_line.txt 2> $O/cfg2b.err || true
unset ADELL_BENCH_ONLY
unset ADELL_WGRAD_STREAM
cd $R
python3 tools/trace_stats.py $O/cfg2b > $O/backbone_unet_kernel_stats.txt 2>&1 || true
python3 tools/trace_stats.py $O/unetr > $O/unetr_kernel_stats.txt 2>&1 || true
python3 tools/trace_stats.py $O/ssl > $O/ssl_convnext_kernel_stats.txt 2>&1 || true
python3 tools/trace_stats.py $O/swin > $O/swinunet_kernel_stats.txt 2>&1 || true
# keep the merge small: the raw traces stay on the box
find $O -name "*.csv" -size +3M -delete
ls $O; du -sh $O
