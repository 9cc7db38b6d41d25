#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <kernel-pattern> "<counters>" -- <command...>   (run on the GPU box from the repo root)
tag=$1; pat=$2; ctrs=$3; shift 4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$tag -o p --output-format csv -- "$@" > /tmp/pmc_$tag.log 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $pat
