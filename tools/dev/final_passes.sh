#!/bin/bash
# dev helper (GPU box): everything the committed profiles/ of a round come from, in the order they depend on each other — the rocprofv3
# passes (config 2, config-5 cell, the "next" rows), their summaries (into profiles/ of this copy, so that the bench lines that follow
# quote the counters of their own sources, and into gpurun_out/profiles_out/ for the trip home), the bench lines of the four workloads,
# the GPU test-suite and the smoke test.   [WITH_CONFIG5=1] [TESTS=tests/test_gpu_asmpw.py] tools/dev/final_passes.sh r04
# (whole config 5 holds 19 volumes on the host and takes six minutes: off unless asked for)
TAG=${1:-r04}
R=$(pwd); P=$R/gpurun_out/profiles_out; F=$R/gpurun_out/final_lines
rm -rf $R/gpurun_out/prof_${TAG}* $P $F; mkdir -p $P $F
bash tools/dev/profile_bench.sh $TAG > gpurun_out/pb.log 2>&1
bash tools/dev/profile_workload.sh $TAG config5_cell > gpurun_out/pw.log 2>&1
bash tools/dev/profile_next_rows.sh $TAG > gpurun_out/pn.log 2>&1
cd $R
O=gpurun_out/prof_$TAG
python profiles/summarize.py $TAG $O/k $O/FETCH_SIZE $O/WRITE_SIZE 2 > /dev/null
python profiles/summarize_sq.py $TAG 2 $O/sq1 $O/sq2 $O/sq3 $O/sq4 > /dev/null
rm -f /tmp/align_$TAG.s; python profiles/finish.py $TAG > /dev/null
cp $P/* profiles/
for f in kernel_stats.csv hbm_counters.md hbm_traffic.json instruction_mix.md instruction_mix.json pmc_source.json agent_info.csv; do cp profiles/${TAG}_$f $P/ 2>/dev/null; done
(timeout 900 python bench.py > $F/config2.json 2> $F/config2.err)
(timeout 600 python bench.py --workload config3 --steps 2 --warmup 1 > $F/config3.json 2> $F/config3.err)
(timeout 600 python bench.py --workload config5_cell --steps 2 --warmup 1 > $F/config5_cell.json 2> $F/config5_cell.err)
(timeout 900 python bench.py --workload config4 > $F/config4.json 2> $F/config4.err)
if [ -n "$WITH_CONFIG5" ]; then (timeout 2200 python bench.py --workload config5 --steps 1 --warmup 0 > $F/config5.json 2> $F/config5.err); fi
wc -c $F/*.json
if [ -n "$WITH_SWEEPS" ]; then      # the randomised parity sweeps against the oracle (profiles/rNN_parity_sweeps.md quotes their last lines)
  (SEED=4501 N=60 timeout 1500 python tools/dev/parity_sweep.py > gpurun_out/sweep_candidates_$TAG.log 2>&1; tail -2 gpurun_out/sweep_candidates_$TAG.log)
  (SEED=4503 N=160 JOBS=4000 timeout 1500 python tools/dev/align_sweep.py > gpurun_out/sweep_extensions_$TAG.log 2>&1; tail -2 gpurun_out/sweep_extensions_$TAG.log)
fi
timeout 1800 python -m pytest ${TESTS:-tests} -q -m gpu -x > gpurun_out/gputest_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/gputest_$TAG.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
