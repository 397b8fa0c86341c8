#!/bin/bash
# dev helper (GPU box): index build timing (production lib), section clocks (ixstats lib when present), then the index parity tests
timeout 300 python tools/dev/idx_time.py 2>&1 | tail -16
if [ -f mecat_amd/lib/libmecat_hip_ixstats.so ] && [ -z "$NOSTATS" ]; then
  MECAT_HIP_LIB=$PWD/mecat_amd/lib/libmecat_hip_ixstats.so timeout 300 python tools/dev/idx_time.py 2>&1 | grep "ix stats" | tail -16
fi
if [ -z "$NOTEST" ]; then timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_parity.py -k index tests/test_gpu_fullsize.py -k index 2>&1 | tail -3; fi
