for lib in tools/probes/ab/libfd_before.so futuredet_amd/libfuturedet_hip.so; do
for c in "--config 3" "--config 5"; do for fl in 1 4; do FD_CONV_STRIP=-1 FD_LIB_PATH=$lib python bench.py $c --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-host-leg --inflight $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib nostrip bench $c inflight $fl', d['value'], d['ms_per_step'])"; done; done
done
