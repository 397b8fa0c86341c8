#!/bin/bash
# dev helper (GPU box): seeding of one nanopore-mode cell (tools/dev/ont_cell.py) with the production library and every libmecat_hip_var*.so
for l in mecat_amd/lib/libmecat_hip_var*.so mecat_amd/lib/libmecat_hip.so; do
  [ -f "$l" ] || continue
  echo "== $l"; MECAT_HIP_LIB=$PWD/$l N=${N:-100000} timeout 600 python tools/dev/ont_cell.py 2>&1 | grep -E "seed |sha256|launches|rror" | head -${LINES_PER:-12}
done
