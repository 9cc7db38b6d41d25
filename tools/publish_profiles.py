"""Copies one tools/pmc_round.sh collection (gpurun_out/<tag>_*) to profiles/<round>_* -- the tracked copies the documents
cite -- and writes the conv-only excerpts of the three PMC passes (the raw counter CSVs hold every dispatch of the run).
usage: python tools/publish_profiles.py <tag> <round>      e.g.  r2g round2"""
import collections
import csv
import os
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
for name in ("bench_default.json", "bench_profiled.json", "bench_serial_profiled.json", "kernel_stats.csv", "serial_kernel_stats.csv",
             "serial_step_summary.txt", "serial_step_sequence.txt", "pmc_summary.txt", "pmc.json"):
    p = os.path.join(src, "%s_%s" % (tag, name))
    if os.path.isfile(p):
        shutil.copyfile(p, os.path.join(dst, "%s_%s" % (rnd, name)))
    else:
        print("missing", p)
# stamp the PMC record with the commit it is published at (the GPU box has no .git); several tags may be merged into one file
pj = os.path.join(dst, "%s_pmc.json" % rnd)
if os.path.isfile(pj):
    import json
    import subprocess

    d = json.load(open(pj))
    extra = sys.argv[3:]  # further tags whose records (same kernel sources) are merged in
    for t in extra:
        q = os.path.join(src, "%s_pmc.json" % t)
        if os.path.isfile(q):
            e = json.load(open(q))
            if e.get("_meta", {}).get("csrc_sha16") == d.get("_meta", {}).get("csrc_sha16"):
                d.update({k: v for k, v in e.items() if k != "_meta"})
            else:
                print("not merged (different kernel sources):", q)
    try:
        d.setdefault("_meta", {})["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=root).decode().strip()
    except Exception:
        pass
    json.dump(d, open(pj, "w"), indent=1)
if os.path.isfile(os.path.join(src, "parity_report.txt")):
    shutil.copyfile(os.path.join(src, "parity_report.txt"), os.path.join(dst, "%s_parity_report.txt" % rnd))


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    return k.replace(", ", ",")


for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
    p = os.path.join(src, "%s_pmc_%s.csv" % (tag, counter))
    if not os.path.isfile(p):
        print("missing", p)
        continue
    rows = list(csv.DictReader(open(p)))
    order = sorted({(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in rows})  # one entry per dispatch, in dispatch order
    starts = [d for i, (d, k) in enumerate(order) if "vox_init" in k and (i == 0 or "vox_" not in order[i - 1][1])]  # first voxelizer launch of a pass
    lo, hi = (starts[-3], starts[-1]) if len(starts) >= 3 else (0, 1 << 62)
    per = collections.OrderedDict()
    for r in rows:
        d = int(r["Dispatch_Id"])
        k = short(r["Kernel_Name"])
        if not (lo <= d < hi) or not (k.startswith("spconv") or k.startswith("conv2d")):
            continue
        e = per.setdefault((d, k, r["Counter_Name"]), [0.0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3])
        e[0] += float(r["Counter_Value"])
    with open(os.path.join(dst, "%s_pmc_%s_convs.txt" % (rnd, counter)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc %s ... -- python bench.py --inflight 1 --graph 0 --steps 4 --warmup 2 --no-cpu-baseline "
                "--no-host-leg; sparse / dense conv dispatches of the last two passes (counter summed over its XCC / SE rows)\n" % counter)
        f.write("dispatch kernel counter value duration_us\n")
        for (d, k, c), (v, us) in per.items():
            f.write("%d %s %s %f %.1f\n" % (d, k, c, v, us))
print("published", tag, "->", rnd)
