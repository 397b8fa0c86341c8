#!/bin/bash
# dev helper (GPU box): SQ counters of dw_extend2 for one bench pass
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-e2e"
rm -rf /tmp/pmc_dw*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA" "SQ_INSTS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_dw$i -- $CMD > /tmp/pmc_dw$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmc_dw*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("dw_extend"):
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    print(k, {n: "%.3e" % v for n, v in sorted(d.items())})
PY
