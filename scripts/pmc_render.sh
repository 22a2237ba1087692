#!/bin/bash
# usage: scripts/pmc_render.sh OUTDIR [spp]   (RT_AMD_LIB selects the library variant)
# Three counter passes over one 1080p render; per-kernel sums via scripts/pmc_summary.py.
OUT=$1; SPP=${2:-16}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P -d $R/gpurun_out/$OUT/p$i -o p$i --output-format csv -- python3 $R/scripts/prof_render.py $SPP > $R/gpurun_out/$OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/$OUT/p$i.log; exit 1; }
done
python3 $R/scripts/pmc_summary.py $(find $R/gpurun_out/$OUT -name "*counter_collection.csv") > $R/gpurun_out/$OUT/summary.txt
sed -n '/k_extend/,/k_generate/p' $R/gpurun_out/$OUT/summary.txt
