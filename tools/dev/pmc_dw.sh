#!/bin/bash
# dev helper (GPU box): one rocprofv3 --pmc pass of the hot path, per-launch sums of the given counters for dw_extend2
# usage: bash tools/dev/pmc_dw.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
SET=${1:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"}
R=$(pwd); O=$R/gpurun_out/pmc_dw
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-e2e > $O/log 2>&1
python - <<P
import glob, pandas as pd
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)
d = pd.concat(pd.read_csv(x) for x in f)
d = d[d["Kernel_Name"].str.contains("dw_extend2")]
n = d["Dispatch_Id"].nunique()
print("launches", n)
print((d.groupby("Counter_Name")["Counter_Value"].sum() / n).to_string())
P
find $O -name "*.db" -delete 2>/dev/null
