#!/bin/bash
# usage (on the GPU box, from the repo root): tools/quick_step_profile.sh <tag> [bench args...]
# One serial bench run under rocprofv3 --kernel-trace; prints the per-kernel summary of one graph-replayed sweep (pass 12 of the trace).
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/q_$tag -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py "$@" --inflight 1 --no-cpu-baseline --no-host-leg --no-also --steps 8 --reps 1 > $out/${tag}_bench_serial.json 2> /tmp/q_$tag.err
f=$(find /tmp/q_$tag -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $f 1 ${SERIAL_PASS:-12} > $out/${tag}_serial_step_summary.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_sequence.py $f ${SERIAL_PASS:-12} > $out/${tag}_serial_step_sequence.txt 2>&1
cat $out/${tag}_serial_step_summary.txt
