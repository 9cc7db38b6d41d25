"""Ordered kernel sequence of the last full step of bench.py from a rocprofv3 --kernel-trace CSV (a step starts at its first voxelizer launch)."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import pass_starts  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = pass_starts(rows)
k = int(sys.argv[2]) if len(sys.argv) > 2 else None  # k-th forward pass of a serial run, else the last full one
a, b = (starts[k], starts[k + 1]) if k is not None else (starts[-2], starts[-1])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    print("%9.1f  %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n))
