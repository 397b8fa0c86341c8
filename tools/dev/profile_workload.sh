#!/bin/bash
# dev helper (GPU box): the rocprofv3 passes behind profiles/rNN_<workload>_*: kernel trace + stats, HBM counters (FETCH_SIZE and
# WRITE_SIZE in separate passes), SQ instruction counters — all of `python bench.py --workload <workload> --steps 1 --warmup 1 --no-cpu`
# (three passes of the hot path per run: warm-up, step, and the untimed pass that brings the cell tables to the host).  Summaries land in gpurun_out/profiles_out/ (copy them to profiles/ and commit).
#   tools/dev/profile_workload.sh r04 config5_cell
TAG=${1:-r04}; WL=${2:-config5_cell}
R=$(pwd); O=$R/gpurun_out/prof_${TAG}_$WL
rm -rf $O; mkdir -p $O $R/gpurun_out/profiles_out
export MECAT_BENCH_VOLCACHE=/dev/shm/mecat_volcache
cd /tmp && export TMPDIR=/tmp
FLAGS="--workload $WL --steps 1 --warmup 1 --no-cpu"
CMD="python $R/bench.py $FLAGS"
timeout 900 $CMD > $O/plain.json 2> $O/plain.err        # fills the volume cache, and the bench line of the same sources outside rocprof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- $CMD > $O/k.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/$c -- $CMD > $O/$c.log 2>&1
done
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/sq$i -- $CMD > $O/sq$i.log 2>&1
done
cd $R
export PROFILES_OUT=$R/gpurun_out/profiles_out PROFILE_CMD="python bench.py $FLAGS" PROFILE_LABEL="$WL"
python profiles/summarize.py ${TAG}_$WL $O/k $O/FETCH_SIZE $O/WRITE_SIZE 3      # warm-up + step + the untimed pass that collects the cell tables
python profiles/summarize_sq.py ${TAG}_$WL 3 $O/sq1 $O/sq2 $O/sq3
python - <<PY
import json, sys
sys.path.insert(0, "$R")
import bench
json.dump({"kernel_source_digest": bench.src_digest(), "command": "python bench.py $FLAGS",
           "tool": "rocprofv3 --kernel-trace --stats / --pmc (separate passes), tools/dev/profile_workload.sh"},
          open("$R/gpurun_out/profiles_out/${TAG}_${WL}_pmc_source.json", "w"), indent=1, sort_keys=True)
PY
cp $O/plain.json $R/gpurun_out/profiles_out/${TAG}_${WL}_bench.json
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +20M -delete 2>/dev/null
rm -rf /dev/shm/mecat_volcache
du -sh $O; ls $R/gpurun_out/profiles_out
