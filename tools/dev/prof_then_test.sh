#!/bin/bash
# development: phase profile of seed_strand with the -DFS_PROF build (mecat_amd/lib/libmecat_hip_prof.so), then the parity tests with the default build
cd "$(dirname "$0")/../.."
cp mecat_amd/lib/libmecat_hip.so /tmp/orig.so
cp mecat_amd/lib/libmecat_hip_prof.so mecat_amd/lib/libmecat_hip.so
timeout 300 python tools/dev/seed_prof.py 2>&1 | grep -E "FS_PROF|seed" | tail -17
cp /tmp/orig.so mecat_amd/lib/libmecat_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
timeout 300 python tools/dev/seed_prof.py 2>&1 | grep -E "seed" | tail -5
