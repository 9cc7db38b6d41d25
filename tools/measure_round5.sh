out=gpurun_out/r5f; mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-also --steps 50 --reps 5 "$@" > $out/$name.json 2> $out/$name.err; tail -1 $out/$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$name', d['value'], d['ms_per_step'], 'h2h', d.get('value_host_to_host'), 'lat', d.get('latency_ms_inflight1'), 'spconv', r.get('spconv_ms_per_step'), 'frac', r.get('frac'), 'min/max', (d.get('repetitions') or {}).get('value_min'), (d.get('repetitions') or {}).get('value_max'))"; }
run default
run batch2 --batch 2
run batch2_if2 --batch 2 --inflight 2
run inflight3 --inflight 3
run inflight6 --inflight 6
run default_b
run config3 --config 3
run config3_b2 --config 3 --batch 2
run config4 --config 4 --no-host-leg
run config5 --config 5 --no-host-leg
