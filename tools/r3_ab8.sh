set -e
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do for lib in libadellhip_lb3.so libadellhip_lb4.so libadellhip.so; do
  ADELL_HIP_LIBRARY=$R/adell_mri_amd/$lib timeout -k 10 300 python tools/small_trio_time.py 128 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], d['cin2_fwd'], d['cin2_fwd_f16x3'])"
done; done
