#!/bin/bash
# dev helper (GPU box): the rocprofv3 passes behind profiles/r01_*: kernel trace + stats, HBM counters, SQ instruction counters,
# all of `python bench.py --steps 1 --warmup 1 --no-cpu` (two passes of the hot path per run).  Outputs under gpurun_out/prof_*.
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k -- $CMD > $O/prof_k.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/prof_$c -- $CMD > $O/prof_$c.log 2>&1
done
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/prof_sq$i -- $CMD > $O/prof_sq$i.log 2>&1
done
ls $O | grep prof_ | head -20
