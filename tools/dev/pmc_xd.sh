#!/bin/bash
# dev helper: SQ counters for the X-drop kernels of tools/dev/bench_xdrop.py (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc$i -- python /root/repo/tests/${PMC_SCRIPT:-bench_xdrop.py} > /tmp/pmc$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        import os
        if os.environ.get("PMC_FILTER", "xd_") in k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    print(k)
    for n, v in sorted(d.items()):
        print("   %-22s %.4g" % (n, v))
PY
