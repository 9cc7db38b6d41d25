#!/bin/bash
# Tuning build: the product sources with -DFD_V2_TRACE (phase timeline of the fp32 sparse conv) -> tools/probes/libfd_trace.so
# Use with FD_LIB_PATH=tools/probes/libfd_trace.so python tools/spconv_trace.py
set -e
cd "$(dirname "$0")/../.."
srcs="fd_error fd_voxelize fd_index fd_spconv fd_spconv_v2 fd_spconv_c32 fd_spconv_bf16 fd_densify fd_conv2d fd_conv2d_f32 fd_conv2d_wino fd_conv2d_wino_pc fd_decode fd_sweeps fd_pillars fd_forecast"
objs=""
mkdir -p tools/probes/_obj
for s in $srcs; do
  extra=""
  case $s in fd_decode|fd_sweeps|fd_forecast|fd_voxelize) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DFD_V2_TRACE $extra -c futuredet_amd/csrc/$s.hip -o tools/probes/_obj/$s.o &
  objs="$objs tools/probes/_obj/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfd_trace.so $objs
echo built tools/probes/libfd_trace.so
