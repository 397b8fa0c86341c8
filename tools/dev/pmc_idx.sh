#!/bin/bash
# dev helper (GPU box): SQ counters of the index-build kernels (tools/dev/idx_time.py), summarised per kernel
R=$(pwd); O=$R/gpurun_out/pmc_idx; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/s$i -- python $R/tools/dev/idx_time.py > $O/s$i.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("/root/repo/gpurun_out/pmc_idx/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "ix_" not in k and "idx_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k[:60], {c: "%.3g" % v for c, v in sorted(d.items())})
P
find $O -name "*.db" -delete 2>/dev/null
