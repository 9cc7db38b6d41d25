#!/bin/bash
# Round-5 evidence set (run on the GPU box from the repo root; the tuning builds must exist: see the build lines below, run where hipcc is).
#   tools/probes/build_exp.sh fd_spconv_bf16win trace -DFD_WIN_TRACE; tools/probes/build_exp.sh fd_decode trace -DFD_DEC_TRACE
#   tools/probes/build_exp.sh fd_spconv_c32 img -DFD_SKELETON_IMAGE; hipcc --offload-arch=gfx950 -O3 -o tools/probes/gather_probe tools/probes/gather_probe.hip
out=gpurun_out/r5ev; mkdir -p $out
nf() { grep -v amdgpu.ids; }
{ echo "# python tools/spconv_bench.py --dtype bf16 --levels 1,2,3 --win 0,-1   (win=0: LDS-window kernel where it applies, -1: RING / RESIDENT kernels)";
  python tools/spconv_bench.py --dtype bf16 --levels 1,2,3 --win 0,-1 2>&1 | nf;
  echo "# four-wave window shapes (bf16_nw = 4), row groups 1..4"; python tools/spconv_bench.py --dtype bf16 --levels 2,3 --win 0 --exp 4 --rg 1,2,3,4 2>&1 | nf; } > $out/bf16win_bench.txt
{ echo "# FD_LIB_PATH=tools/probes/libfd_fd_spconv_bf16win_trace.so python tools/bf16win_trace.py 0   (tuning build -DFD_WIN_TRACE: ~25 % slower than the product)";
  FD_LIB_PATH=tools/probes/libfd_fd_spconv_bf16win_trace.so python tools/bf16win_trace.py 0 2>&1 | nf; } > $out/bf16win_phase_trace.txt
python tools/window_stats.py 2>&1 | nf > $out/window_stats.txt
tools/probes/gather_probe 16 > $out/gather_probe.txt 2>&1
{ FD_LIB_PATH=tools/probes/libfd_fd_decode_trace.so python tools/decode_trace.py 2>&1 | nf; } > $out/decode_trace.txt
{ FD_LIB_PATH=tools/probes/libfd_fd_spconv_c32_img.so python tools/skeleton_image_bench.py 1 2>&1 | nf; } > $out/skeleton_image_raw.txt
for w in 0 -1; do FD_BF16_WIN=$w python bench.py --config 3 --no-cpu-baseline --steps 50 --reps 5 > $out/config3_win$w.json 2>/dev/null; done
python - <<PY > $out/config3_window_ab.txt
import json
for w in ("0", "-1"):
    d = json.loads(open("$out/config3_win%s.json" % w).read().strip().splitlines()[-1]); r = d["roofline"]
    print("FD_BF16_WIN=%s: %.1f sweeps/s (%.4f ms/step, repetitions %s), latency one pass in flight %.3f ms, sparse conv %.3f ms per step, frac %.4f" % (
        w, d["value"], d["ms_per_step"], d["repetitions"]["ms_per_step_each"], d["latency_ms_inflight1"], r["spconv_ms_per_step"], r["frac"]))
PY
bash tools/measure_round5.sh > $out/measure_round.txt 2>&1
ls -la $out
