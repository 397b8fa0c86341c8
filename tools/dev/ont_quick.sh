#!/bin/bash
# dev helper (GPU box): one nanopore-mode cell (tools/dev/ont_cell.py): production library timings + result hashes, then section clocks (wfprof)
N=${N:-100000} timeout 600 python tools/dev/ont_cell.py 2>&1 | grep -E "seed |sha256|launches|rror" | head -12
if [ -f mecat_amd/lib/libmecat_hip_wfprof.so ] && [ -z "$NOPROF" ]; then
  MECAT_HIP_LIB=$PWD/mecat_amd/lib/libmecat_hip_wfprof.so N=${N:-100000} timeout 600 python tools/dev/ont_cell.py 2>&1 | grep -E "wf prof" | tail -8
fi
