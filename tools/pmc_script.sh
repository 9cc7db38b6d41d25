#!/bin/bash
# usage (GPU box, repo root): tools/pmc_script.sh <tag> <script.py> [script args]   -- SQ / TA / TCP counters of the sparse-conv kernels
# (several --pmc passes; never combined with tracing domains other than --kernel-trace); summary -> gpurun_out/<tag>_spconv_pmc.txt
tag=$1; shift; SCRIPT=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
            "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE" \
            "TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pass -d /tmp/sp_${tag}_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/$SCRIPT "$@" > /tmp/sp_${tag}_$i.log 2>&1
  f=$(find /tmp/sp_${tag}_$i -name "*counter_collection.csv" | head -1)
  cp $f $out/${tag}_spconv_pmc_$i.csv 2>/dev/null || { echo "pass $i failed"; tail -5 /tmp/sp_${tag}_$i.log; }
done
python - <<PY
import collections, csv, glob
acc = collections.OrderedDict()
for f in sorted(glob.glob("$out/${tag}_spconv_pmc_*.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        if "spconv_" not in k:
            continue
        k = k.split("(")[0]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    for k in per:
        for c, v in per[k].items():
            acc.setdefault(k, collections.OrderedDict())[c] = v / max(len(cnt[k]), 1)
with open("$out/${tag}_spconv_pmc.txt", "w") as o:
    for k, d in acc.items():
        o.write(k + "\n")
        for c, v in d.items():
            o.write("    %-36s %16.0f\n" % (c, v))
print(open("$out/${tag}_spconv_pmc.txt").read())
PY
