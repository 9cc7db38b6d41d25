#!/bin/bash
# rocprofv3 kernel durations of the fp32 16 -> 16 layer (spconv_f32_res16) per gather ring depth: eager event timing of this kernel is
# host-bound (~12 us per ctypes launch), so the figure comes from the kernel trace.  usage (GPU box): tools/res16_depth_prof.sh
cd /tmp && export TMPDIR=/tmp
for d in 0 8 12 16; do
  rm -rf /tmp/rq_$d
  rocprofv3 --kernel-trace --stats -d /tmp/rq_$d -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/spconv_bench.py --levels 0 --modes uniform --iters 30 --depth $d > /dev/null 2>&1
  f=$(find /tmp/rq_$d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$d" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "res16" in r["Name"]:
        print("depth=%s %s calls %s avg %.1f us min %.1f us" % (sys.argv[2], r["Name"].split("(")[1][:45] if False else r["Name"][29:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
