#!/bin/bash
# dev helper (GPU box): the rocprofv3 passes behind profiles/rNN_*: kernel trace + stats, HBM counters (FETCH_SIZE / WRITE_SIZE in
# separate passes), SQ instruction counters — all of `python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --no-e2e`
# (two passes of the hot path per run).  Raw outputs under gpurun_out/prof_<tag>/; summarise with profiles/summarize*.py.
TAG=${1:-r02}
R=$(pwd); O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-extras --no-e2e"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- $CMD > $O/k.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -- $CMD > $O/$c.log 2>&1
done
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/sq$i -- $CMD > $O/sq$i.log 2>&1
done
# keep only what the summaries read
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +20M -delete 2>/dev/null
du -sh $O; ls $O
