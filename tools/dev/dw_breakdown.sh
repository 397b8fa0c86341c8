#!/bin/bash
# dev helper (GPU box): section clocks + row statistics of dw_extend2 (the -DMECAT_DW_STATS build, `make dwstats` in the container)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
MECAT_HIP_LIB=$PWD/mecat_amd/lib/libmecat_hip_dwstats.so N=${N:-20000} timeout 600 python tools/dev/dw_breakdown.py > gpurun_out/dw_breakdown.md 2> gpurun_out/dw_breakdown.err
tail -40 gpurun_out/dw_breakdown.md; tail -5 gpurun_out/dw_breakdown.err
