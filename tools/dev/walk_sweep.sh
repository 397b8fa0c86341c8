#!/bin/bash
# development: time seed_strand's phases for the walk pipeline variants built as mecat_amd/lib/libmecat_hip_<cfg>.so
cd "$(dirname "$0")/../.."
cp mecat_amd/lib/libmecat_hip.so /tmp/orig.so
for f in mecat_amd/lib/libmecat_hip_*.so; do
  cp $f mecat_amd/lib/libmecat_hip.so
  echo "== $f"
  timeout 300 python tools/dev/seed_prof.py 2>&1 | grep -E "FS_PROF phase (1|4):|seed_strand|^seed" | tail -5
done
cp /tmp/orig.so mecat_amd/lib/libmecat_hip.so
