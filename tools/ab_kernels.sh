#!/bin/bash
# A/B of two builds of the library on the same box: rocprofv3 kernel durations of the sparse conv kernels over a short bench run.
# usage: tools/ab_kernels.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf /tmp/ab_$tag
  FD_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ab_$tag -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline > /tmp/ab_$tag.json 2>/tmp/ab_$tag.err < /dev/null
  f=$(ls /tmp/ab_$tag/*/k_kernel_stats.csv /tmp/ab_$tag/k_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $lib  $(tail -1 /tmp/ab_$tag.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null)"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r.get("Name", "")
    if "spconv_f32" in n or "conv2d_wino" in n:
        print("   %-60s calls %5s avg %9.1f us" % (n.replace("(anonymous namespace)::", "")[:60], r.get("Calls"), float(r.get("AverageNs", 0)) / 1e3))
PY
  else echo "no stats file"; ls /tmp/ab_$tag | head; fi
done
