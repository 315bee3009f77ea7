set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02e
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02e/stats -- python3 $R/bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-fp32 > $R/gpurun_out/r02e/bench_line.json 2> $R/gpurun_out/r02e/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02e/pmcF -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 > $R/gpurun_out/r02e/pmcF.json 2> $R/gpurun_out/r02e/pmcF.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02e/pmcW -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 > $R/gpurun_out/r02e/pmcW.json 2> $R/gpurun_out/r02e/pmcW.err
cd $R
python3 bench.py > gpurun_out/r02e/bench_plain.json 2> gpurun_out/r02e/bench_plain.err
ls gpurun_out/r02e; find gpurun_out/r02e -name "*.csv" | head; du -sh gpurun_out/r02e
