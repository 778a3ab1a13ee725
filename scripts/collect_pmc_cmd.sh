#!/bin/bash
# Runs ON THE GPU BOX.  SQ / TCC counters of an arbitrary command, one small counter set per rocprofv3
# pass (--pmc with --kernel-trace only: the combination gpurun allows).  usage: collect_pmc_cmd.sh <outdir> <kernel-regex> -- <cmd ...>
out=$1; pat=$2; shift 3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o pmc -- "$@" > $out/p$i.out 2> $out/p$i.err || echo "pass $i failed/timeout"
done
python scripts/pmc_summary.py $out | grep -E "^==|$pat" > $out/summary.txt
cat $out/summary.txt
