#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_round.sh <tag> [bench args...]
# Collects, for the same bench command: (1) rocprofv3 --kernel-trace --stats of the default run, (2) three separate --pmc
# passes (FETCH_SIZE / WRITE_SIZE / MFMA-busy) of a short serial run, and writes summaries to gpurun_out/<tag>_*.
# PMC passes never combine with the hip/hsa/memory-copy trace domains.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-also"
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt --output-format csv -- $B "$@" > $out/${tag}_bench_profiled.json 2> /tmp/kt_$tag.err
cp $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
# one step in isolation: a serial run (--inflight 1).  Forward passes of the trace: 1 set-up + 2 StaticStep warm-up + 1 pre-capture run,
# 5 warm-up steps, then the timed steps: pass 12 = the fourth timed step, one whole-sweep hipGraph replay (the fp32 default)
rocprofv3 --kernel-trace --stats -d /tmp/kts_$tag -o kt --output-format csv -- $B "$@" --inflight 1 --no-cpu-baseline --no-host-leg > $out/${tag}_bench_serial_profiled.json 2> /tmp/kts_$tag.err
cp $(find /tmp/kts_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_serial_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/kts_$tag -name "*kernel_trace.csv" | head -1) 1 ${SERIAL_PASS:-12} > $out/${tag}_serial_step_summary.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_sequence.py $(find /tmp/kts_$tag -name "*kernel_trace.csv" | head -1) ${SERIAL_PASS:-12} > $out/${tag}_serial_step_sequence.txt 2>&1
short="--inflight 1 --graph 0 --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg"  # eager launches: the same kernels, one dispatch each
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_${tag}_$name -o p --output-format csv -- $B "$@" $short > /tmp/pmc_${tag}_$name.log 2>&1
  f=$(find /tmp/pmc_${tag}_$name -name "*counter_collection.csv" | head -1)
  cp $f $out/${tag}_pmc_$name.csv 2>/dev/null || echo "no counter file for $name" >&2
done
python $GRAFT_REPO_ROOT/tools/pmc_round_summary.py $out $tag "$@" > $out/${tag}_pmc_summary.txt 2>&1
tail -30 $out/${tag}_pmc_summary.txt
