out=gpurun_out/${1:-r4h}; mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 50 --reps 3 "$@" > $out/$name.json 2> $out/$name.err; tail -1 $out/$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$name', d['value'], d['ms_per_step'], 'h2h', d.get('value_host_to_host'), 'lat', d.get('latency_ms_inflight1'), 'spconv', r.get('spconv_ms_per_step'), 'frac', r.get('frac'), 'min/max', (d.get('repetitions') or {}).get('value_min'), (d.get('repetitions') or {}).get('value_max'))"; }
run default --stage-times
grep stage $out/default.err | tail -2
run inflight2 --inflight 2
run serial --inflight 1 --stage-times
grep stage $out/serial.err | tail -2
run nograph --graph 0
run serial_nograph --graph 0 --inflight 1
timeout 300 python tools/torch_dense_ab.py > $out/torch_dense_ab.txt 2>&1; tail -1 $out/torch_dense_ab.txt
run batch2 --batch 2
run batch8 --batch 8 --no-host-leg
run config3 --config 3
run config3_serial --config 3 --inflight 1 --no-host-leg
run config3_graph --config 3 --graph 1 --no-host-leg
run config4 --config 4 --no-host-leg
run config5 --config 5 --no-host-leg
run n3dtf_bf16 --variant forecast_n3dtf --dtype bf16 --no-host-leg
run pp_fp32 --variant pp_n3dtf --no-host-leg
run pp_bf16 --variant pp_n3dtf --dtype bf16 --no-host-leg
timeout 300 python tools/dense_fp32_bench.py > $out/dense_fp32_bench.txt 2>&1; tail -15 $out/dense_fp32_bench.txt
