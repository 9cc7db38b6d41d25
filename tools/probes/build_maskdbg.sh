#!/bin/bash
# Tuning build that REPRODUCES the round-6 determinism failure and instruments it: fd_decode.hip compiled as it was before the fix -- SLP vectoriser on,
# i.e. ~500 packed-fp32 instructions per IoU evaluation -- with -DFD_MASK_DEBUG (nms_mask evaluates every near pair again from footprints pinned in
# registers and counts / logs the disagreements), linked with the product's other objects -> tools/probes/libfd_maskdbg.so.
#   FD_LIB_PATH=tools/probes/libfd_maskdbg.so python tools/soak_determinism.py bf16 400
#   FD_LIB_PATH=tools/probes/libfd_maskdbg.so python tools/soak_pairs.py 300 bf16 3
# (tools/probes/build_variant.sh fd_decode slp -fslp-vectorize gives the un-instrumented "before" library.)
set -e
cd "$(dirname "$0")/../.."
python futuredet_amd/build.py > /dev/null
mkdir -p tools/probes/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DFD_MASK_DEBUG $FD_EXTRA_DEFS -ffp-contract=off -fslp-vectorize -c futuredet_amd/csrc/fd_decode.hip -o tools/probes/_obj/fd_decode_maskdbg.o
objs=$(ls futuredet_amd/csrc/_obj/*.o | grep -v fd_decode.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfd_maskdbg.so $objs tools/probes/_obj/fd_decode_maskdbg.o
echo built tools/probes/libfd_maskdbg.so
