#!/bin/bash
# dev helper (GPU box): rocprofv3 --pmc passes of the hot path, per-launch sums of the given counters for the kernels matching a pattern
# usage: bash tools/dev/pmc_kernel.sh <kernel regex> "<counters of pass 1>,<counters of pass 2>,.." [bench.py flags]
# (FETCH_SIZE and WRITE_SIZE need a pass each; MECAT_HIP_LIB selects a variant library)
PAT=$1; SETS=${2:-"FETCH_SIZE,WRITE_SIZE"}; shift; shift
R=$(pwd); O=$R/gpurun_out/pmc_k
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
IFS=','; for SET in $SETS; do unset IFS
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-e2e "$@" > $O/log$i 2>&1
done
python - <<P
import glob, pandas as pd
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)
d = pd.concat(pd.read_csv(x) for x in f)
d = d[d["Kernel_Name"].str.contains("$PAT")]
d["k"] = d["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace(r"^void ", "", regex=True)
g = d.groupby(["k", "Counter_Name"])["Counter_Value"].sum().unstack()
print(g.to_string())
P
find $O -name "*.db" -delete 2>/dev/null
