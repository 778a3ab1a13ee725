#!/bin/bash
# Runs ON THE GPU BOX.  SQ / GRBM / TCC counters of the bench command, one small set per pass (only counter
# sets that have run cleanly on this pool; TA_* / TCP_* sets hung a box once and are not used).
tag=${1:-r01}; wl=${2:-c3}
out=gpurun_out/pmc_sq; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline --no-live-traffic --bootstraps 0 --cells 0"
i=0
for set in "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_EA_BUSY GRBM_TC_BUSY" "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o $tag -- $B > /dev/null 2> $out/p$i.err || echo "pass $i failed/timeout"
done
python scripts/pmc_summary.py $out | grep -E "^==|k_em_tile |k_remote_fold |k_reldiff" > $out/${tag}_${wl}_pmc_sq_tcc_summary.txt
cat $out/${tag}_${wl}_pmc_sq_tcc_summary.txt
