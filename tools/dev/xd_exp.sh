#!/bin/bash
# dev: what bounds xd_extend_w?  Variants of the library with parts of the kernel removed (results wrong, timing only)
#   no variant runs the traceback (it would walk bytes that were never written):
#   XD_EXP=3 every store of the row loop   XD_EXP=1 without the script-byte store of the passes   XD_EXP=2 no stores in the row loop
R=$(pwd)
for e in 1 2 3; do
  mkdir -p build/xd$e
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imecat_amd/csrc -DXD_EXP=$e -x hip -c mecat_amd/csrc/xalign.hip -o build/xd$e/xalign.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/xd$e/xalign.o $(ls build/*.o | grep -v xalign.o) -o mecat_amd/lib/libmecat_hip_xd$e.so
done
