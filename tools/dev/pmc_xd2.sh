#!/bin/bash
# dev (GPU box): SQ counters of the X-drop kernels on tools/dev/xd_time.py's candidates -> gpurun_out/pmc_xd.txt
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcx*; i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmcx$i -- python $R/tools/dev/xd_time.py > /tmp/pmcx$i.log 2>&1
done
python3 - > $R/gpurun_out/pmc_xd.txt <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmcx*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "xd_extend" in k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in tot.items():
    print(k)
    for n, v in sorted(d.items()):
        print("   %-22s %.4g per launch (%d launches)" % (n, v / cnt[k][n], cnt[k][n]))
PY
cat $R/gpurun_out/pmc_xd.txt
