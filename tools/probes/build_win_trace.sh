#!/bin/bash
# Tuning build: product objects + fd_spconv_bf16w.hip with -DFD_WIN_TRACE (phase cycle counters) -> tools/probes/libfd_win_trace.so
# Use: FD_LIB_PATH=tools/probes/libfd_win_trace.so python tools/win_trace.py
set -e
cd "$(dirname "$0")/../.."
python futuredet_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DFD_WIN_TRACE -c futuredet_amd/csrc/fd_spconv_bf16w.hip -o tools/probes/_obj/fd_spconv_bf16w_trace.o
objs=$(ls futuredet_amd/csrc/_obj/*.o | grep -v fd_spconv_bf16w.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfd_win_trace.so $objs tools/probes/_obj/fd_spconv_bf16w_trace.o
echo built tools/probes/libfd_win_trace.so
