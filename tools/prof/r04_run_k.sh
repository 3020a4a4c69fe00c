O=gpurun_out
timeout 900 python bench.py --steps 10 --warmup 2 2>$O/r04_k_bench.err | tail -1 > $O/r04_k_bench.json
ZH_L1_PARSE=parallel timeout 300 python bench.py --buffers 512 --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --no-parity-sample 2>/dev/null | tail -1 > $O/r04_k_share512_parallel_parse.json
timeout 300 python bench.py --buffers 512 --steps 10 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/r04_k_share512.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_k_bench.json'))
print('value', d['value'], d['value_parallel_parse'], d['roofline']['traffic'], d['parallel_parse']['roofline_matcher']['traffic'])
for k,v in d['configs'].items(): print(k, v['value'], v['dominant_kernel'], v['frac'], v['traffic'], v['algorithmic_bytes'])
cb=d['cpu_baseline']; print(cb['value'], cb['value_at_min'], cb['cores'], {k:v['both_GiBps_at_avg'] for k,v in cb['other_thread_counts'].items()}, cb['all_cores']['oracle']['compress'], cb['all_cores']['oracle']['uncompress'])
print({k:(v['traffic'], v['frac'], v['launches_per_step']) for k,v in d['roofline_kernels'].items()})
for f in ('share512','share512_parallel_parse'):
    e=json.load(open('gpurun_out/r04_k_%s.json'%f)); print(f, e['value'], e['ms_per_step'], {k:round(v,2) for k,v in e['kernels_ms'].items() if v>0.2})
PY
