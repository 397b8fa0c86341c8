#!/bin/bash
# development: run seed_prof.py with every mecat_amd/lib/libmecat_hip_<variant>.so in turn
cd "$(dirname "$0")/../.."
cp mecat_amd/lib/libmecat_hip.so /tmp/orig.so
for f in mecat_amd/lib/libmecat_hip_*.so; do
  cp $f mecat_amd/lib/libmecat_hip.so
  echo "== $f"
  timeout 300 python tools/dev/seed_prof.py 2>&1 | grep -E "${PAT:-seed_}" | tail -6
done
cp /tmp/orig.so mecat_amd/lib/libmecat_hip.so
