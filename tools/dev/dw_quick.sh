#!/bin/bash
# dev (GPU box): dw / index / seeding correctness + the config-2 line without its CPU legs -> gpurun_out/$1/
T=${1:-dwq}; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4) > gpurun_out/$T/tests.log
(timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --no-e2e > gpurun_out/$T/config2.json 2> gpurun_out/$T/config2.err)
cat gpurun_out/$T/tests.log
python -c "import json;d=json.load(open('gpurun_out/$T/config2.json'));print(d['value'], d['ms_per_step'], d['phase_ms']); k=d['kernel_ms_per_step']; print({a:round(b,2) for a,b in list(k.items())[:14]})"
