set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
true
true
for l in base "" base ""; do ADELL_HIP_LIBRARY=$PWD/adell_mri_amd/libadellhip${l:+_$l}.so timeout -k 10 200 python tools/ab_lib.py --child 24 | tail -1; done
