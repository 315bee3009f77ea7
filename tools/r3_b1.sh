set -e
R=$GRAFT_REPO_ROOT
cd $R
for env in "X=1" "ADELL_NO_ADN_FUSE=1" "X=2"; do
  env $env timeout -k 10 300 python bench.py --batch 1 --steps 20 --no-cpu-baseline --no-fp32 --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$env', d['ms_per_step'], d['median_ms_per_step'])"
done
