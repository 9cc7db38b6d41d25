#!/bin/bash
# rocprofv3 kernel durations of tools/tiles_bench.py: the compressed-rulebook kernels (fd_spconv_tiles.hip) next to the dense-table kernels they
# replace on the same rulebooks (event timing of these launches through ctypes is host-bound).  usage (GPU box): tools/tiles_prof.sh [bench args]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp
rocprofv3 --kernel-trace --stats -d /tmp/tp -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/tiles_bench.py "$@" > /tmp/tp.out 2>&1
grep -v amdgpu /tmp/tp.out | cut -c1-200
f=$(find /tmp/tp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void (anonymous namespace)::", "")
    if n.startswith(("spconv_", "rulebook")):
        print("%-60s calls %5s avg %8.1f us  min %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
