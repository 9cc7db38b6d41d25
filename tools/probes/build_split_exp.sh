#!/bin/bash
# Tuning builds of the split-operand sparse conv: product objects + fd_spconv_split.hip compiled with -DFD_SPLIT_EXP=<mask>
# -> tools/probes/libfd_split_exp<mask>.so.  Use: FD_LIB_PATH=tools/probes/libfd_split_exp3.so python tools/split_bench.py ...
set -e
cd "$(dirname "$0")/../.."
python futuredet_amd/build.py > /dev/null
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DFD_SPLIT_EXP=$m -c futuredet_amd/csrc/fd_spconv_split.hip -o tools/probes/_obj/fd_spconv_split_exp$m.o &
done
wait
for m in "$@"; do
  objs=$(ls futuredet_amd/csrc/_obj/*.o | grep -v fd_spconv_split.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfd_split_exp$m.so $objs tools/probes/_obj/fd_spconv_split_exp$m.o
done
echo built "$@"
