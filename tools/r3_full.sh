set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1 || (tail -60 $O/full_tests.log | cut -c1-300; exit 1)
tail -3 $O/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/full_bench.json 2> $O/full_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3/full_bench.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['hbm']['ms_per_step'], {k:(v['ms_per_step']) for k,v in d['secondary'].items()})
PY
