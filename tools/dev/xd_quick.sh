#!/bin/bash
# dev (GPU box): X-drop correctness + the two timings that matter -> gpurun_out/$1/
T=${1:-xdq}; mkdir -p gpurun_out/$T
(timeout 600 python -m pytest tests/test_gpu_xalign.py -x -q 2>&1 | tail -5) > gpurun_out/$T/test_xalign.log
(timeout 200 python tools/dev/xd_time.py 2>&1 | tail -1) > gpurun_out/$T/xd_time.log
(timeout 500 python bench.py --workload config5_cell --steps 2 --warmup 1 --no-cpu > gpurun_out/$T/c5cell.json 2> gpurun_out/$T/c5cell.err)
cat gpurun_out/$T/test_xalign.log gpurun_out/$T/xd_time.log
python -c "import json;d=json.load(open('gpurun_out/$T/c5cell.json'));print(d['phase_ms'])"
