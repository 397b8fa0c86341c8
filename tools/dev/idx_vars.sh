#!/bin/bash
# dev helper (GPU box): index build timing of the production library and of every mecat_amd/lib/libmecat_hip_var*.so
for l in mecat_amd/lib/libmecat_hip.so mecat_amd/lib/libmecat_hip_var*.so; do
  [ -f "$l" ] || continue
  echo "== $l"; MECAT_HIP_LIB=$PWD/$l timeout 300 python tools/dev/idx_time.py 2>&1 | grep -E "ix_|total|failed" | head -${LINES_PER:-8}
done
if [ -z "$NOTEST" ]; then timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_parity.py -k index tests/test_gpu_fullsize.py -k index 2>&1 | tail -3; fi
