#!/bin/bash
# PMC A/B of the persistent wave-specialised conv instance against the one-brick-per-block one
# (run on the GPU box): tools/pmc_ws.sh <Cin> <Cout> <size>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in ws nows; do
  if [ $mode = ws ]; then export ADELL_IGEMM_WS=1; else unset ADELL_IGEMM_WS; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$mode -- python3 $R/tools/one_conv.py fwd $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_$mode -- python3 $R/tools/one_conv.py fwd $1 $2 $3 > /dev/null 2>&1
  echo "== $mode"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$mode igemm; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc2_$mode igemm
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/pmc_$mode/**/*kernel_trace.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "igemm" in r["Kernel_Name"]:
        print("dur_us", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:60], "vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"))
PY
done
