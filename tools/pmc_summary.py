import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "spconv"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in rows:
    if pat in r["Kernel_Name"]:
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[k]:
            cnt[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in agg.items():
    n = len(cnt[k])
    print(k, "dispatches", n, "avg %.1f us" % (dur[k] / n))
    for c, val in sorted(v.items()):
        print("   %-30s %.4g" % (c, val / n))
