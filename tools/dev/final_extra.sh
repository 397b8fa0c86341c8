#!/bin/bash
# dev helper (GPU box), second half of a round's final evidence (the first half is tools/dev/final_passes.sh): what does not depend on the
# rocprofv3 passes — the randomised parity sweeps against the oracle (uniform and repeat-structured genomes), the simulated multi-GPU shares
# (bench.py --simulate-ranks), the end-to-end legs of the multi-volume configs and, when asked for, whole config 5.
#   [WITH_CONFIG5=1] tools/dev/final_extra.sh r06
TAG=${1:-r06}
R=$(pwd); F=$R/gpurun_out/final_lines; mkdir -p $F
python -c "import bench, json; print(json.dumps({'kernel_source_digest': bench.src_digest()}))" > $F/extra_source.json
(timeout 600 python bench.py --simulate-ranks 2,4,8 --no-cpu --no-e2e --no-extras > $F/sim_config2.json 2> $F/sim_config2.err; grep simulate $F/sim_config2.err)
(timeout 600 python bench.py --workload config3 --simulate-ranks 4,8 --no-cpu > $F/sim_config3.json 2> $F/sim_config3.err; grep simulate $F/sim_config3.err)
(timeout 600 python bench.py --workload config5_cell --simulate-ranks 2,4,8 --no-cpu > $F/sim_config5_cell.json 2> $F/sim_config5_cell.err; grep simulate $F/sim_config5_cell.err)
(timeout 900 python bench.py --workload config3 --e2e > $F/e2e_config3.json 2> $F/e2e_config3.err; grep "\[e2e\]" $F/e2e_config3.err | cut -c1-200)
(timeout 1500 python bench.py --workload config5 --e2e > $F/e2e_config5.json 2> $F/e2e_config5.err; grep "\[e2e\]" $F/e2e_config5.err | cut -c1-200)
if [ -n "$WITH_CONFIG5" ]; then (timeout 2200 python bench.py --workload config5 --steps 1 --warmup 0 > $F/config5.json 2> $F/config5.err; wc -c $F/config5.json); fi
(SEED=4601 N=60 timeout 1500 python tools/dev/parity_sweep.py > gpurun_out/sweep_candidates_$TAG.log 2>&1; tail -3 gpurun_out/sweep_candidates_$TAG.log)
(SEED=4602 N=60 REP=1 timeout 1800 python tools/dev/parity_sweep.py > gpurun_out/sweep_candidates_rep_$TAG.log 2>&1; tail -3 gpurun_out/sweep_candidates_rep_$TAG.log)
(SEED=4603 N=120 JOBS=3000 timeout 1800 python tools/dev/align_sweep.py > gpurun_out/sweep_extensions_$TAG.log 2>&1; tail -2 gpurun_out/sweep_extensions_$TAG.log)
(SEED=4604 N=120 JOBS=3000 REP=1 timeout 1800 python tools/dev/align_sweep.py > gpurun_out/sweep_extensions_rep_$TAG.log 2>&1; tail -2 gpurun_out/sweep_extensions_rep_$TAG.log)
