#!/bin/bash
# Round-6 evidence set (run on the GPU box from the repo root): the driver's command, kernel traces + PMC passes of the fp32 headline and of
# config 3, the serial step of the full pipeline, the next-row kernels alone.  Results land in gpurun_out/r6ev* ; tools/publish_profiles.py and the
# copies at the end of this script move what is cited into profiles/round6_*.
out=gpurun_out; mkdir -p $out
nf() { grep -v amdgpu.ids; }
python bench.py --steps 20 --warmup 5 > $out/r6ev_bench_default.json 2> $out/r6ev_bench_default.err
tools/pmc_round.sh r6ev > $out/r6ev_pmc_round.log 2>&1
tools/pmc_round.sh r6evc3 --config 3 > $out/r6evc3_pmc_round.log 2>&1
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $out/r6evc3_bench_default.json 2> /dev/null
tools/quick_step_profile.sh r6evfull --pipeline full > /dev/null 2>&1
python tools/next_rows_bench.py 2>&1 | nf > $out/r6ev_next_rows_bench.txt
ls -la $out | grep r6ev
