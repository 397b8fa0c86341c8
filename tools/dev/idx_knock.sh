#!/bin/bash
# dev helper (GPU box): timing experiments of the index build's copy-out (ixstats build; results are wrong under a knock-out)
for k in 0 1 2 3; do echo "== MECAT_IX_KNOCK=$k"; MECAT_IX_KNOCK=$k MECAT_HIP_LIB=$PWD/mecat_amd/lib/libmecat_hip_ixknock.so timeout 300 python tools/dev/idx_time.py 2>&1 | grep -E "ix_scatter|ix_fill|total|failed" | tail -5; done
