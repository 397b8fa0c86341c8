#!/bin/bash
# dev helper (GPU box): config 2's seeding kernels (bench.py without its CPU legs) with the production library and every libmecat_hip_var*.so
for l in mecat_amd/lib/libmecat_hip_var*.so mecat_amd/lib/libmecat_hip.so; do
  [ -f "$l" ] || continue
  echo "== $l"; MECAT_HIP_LIB=$PWD/$l timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-extras --no-e2e --stats /tmp/seed_c2_stats.json > /dev/null 2>&1; python -c "
import json
d=json.load(open('/tmp/seed_c2_stats.json')); k=d['kernels']; l=d['line']
print(l['candidates'], round(l['ms_per_step'],1), l['phase_ms'], {a:round(b['total_ms']/2,2) for a,b in k.items() if a.startswith('seed')})"
done
