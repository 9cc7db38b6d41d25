"""Summarise a rocprofv3 --kernel-trace CSV for the LAST step of bench.py (a step starts at its first voxelizer launch)."""
import csv
import sys
from collections import OrderedDict


def pass_starts(rows):
    """indices of the rows that start a forward pass: the first kernel of a voxelizer call (vox_init) whose predecessor is not a voxelizer
    kernel -- a pass of B clouds runs B voxelizer calls back to back"""
    out = []
    for i, r in enumerate(rows):
        if "vox_init" in r["Kernel_Name"] and (i == 0 or "vox_" not in rows[i - 1]["Kernel_Name"]):
            out.append(i)
    return out



def main(path, steps_back=1, step_index=None):
    """step_index = k: the k-th forward pass of the trace (0-based, counted by its first voxelizer launch; use with a serial
    --inflight 1 run: passes in flight on two streams interleave in time).  Otherwise the steps_back passes before the last."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = pass_starts(rows)
    # full steps only: from the (steps_back+1)-th last step start up to the start of the last step (the trailing
    # part of the trace also holds bench.py's post-loop pair counting, which is not part of a step)
    if step_index is not None:
        first, last, steps_back = starts[step_index], starts[step_index + 1], 1
    else:
        first, last = starts[-1 - steps_back], starts[-1]
    sel = rows[first:last]
    agg = OrderedDict()
    for r in sel:
        n = r["Kernel_Name"]
        n = n.replace("void ", "").replace("(anonymous namespace)::", "")[:70]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(n, [0.0, 0])
        a[0] += d
        a[1] += 1
    span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
    busy = sum(a[0] for a in agg.values())
    print("last %d step(s): %d kernels, span %.1f us, busy %.1f us (%.1f%%)" % (steps_back, len(sel), span, busy, 100 * busy / span))
    for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        print("%9.1f us %4d x  %s" % (d, c, n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else None)
