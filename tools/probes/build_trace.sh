#!/bin/bash
# Tuning build: ALL product sources with -DFD_V2_TRACE (phase timelines of the fp32 sparse conv, the Winograd and the bf16 dense kernels)
# -> tools/probes/libfd_trace.so.  Use with FD_LIB_PATH=tools/probes/libfd_trace.so python tools/spconv_trace.py
# (one source only: tools/probes/build_exp.sh <src> <tag> -DFD_V2_TRACE)
set -e
cd "$(dirname "$0")/../.."
srcs=$(python -c "import sys; sys.path.insert(0, 'futuredet_amd'); import build; print(' '.join(s[:-4] for s in build.SOURCES))")
objs=""
mkdir -p tools/probes/_obj
for s in $srcs; do
  extra=$(python -c "import sys; sys.path.insert(0, 'futuredet_amd'); import build; print(' '.join(build.EXTRA.get('$s.hip', [])))")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DFD_V2_TRACE $extra -c futuredet_amd/csrc/$s.hip -o tools/probes/_obj/$s.o &
  objs="$objs tools/probes/_obj/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfd_trace.so $objs
echo built tools/probes/libfd_trace.so
