O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_e_pytest_gpu.log 2>&1; tail -2 $O/r04_e_pytest_gpu.log
timeout 300 python tools/kprof.py --l1-parse 1 --buffers 1024 > $O/r04_e_kprof_contract.txt 2>&1; grep -A10 "kernel ms\|== zh_l1p_match\|== zh_huffman_kernel" $O/r04_e_kprof_contract.txt | head -34
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>$O/r04_e_bench.err | tail -1 > $O/r04_e_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_e_bench.json'))
print('value', d['value'], 'pp', d['value_parallel_parse'], d['parallel_parse']['size_vs_exact_parse'], d['kernels_ms'])
print('pp kernels', d['parallel_parse']['kernels_ms'])
for k,v in d['configs'].items(): print(k, v['value'], v['ms_per_step'], v['ratio'], v['kernels_ms'] if k.startswith('c2') else v['dominant_kernel_ms'])
PY
timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1; ZH_L1_PARSE=parallel timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1
for sl in 512 768; do ZH_L1P_SLOTS=$sl ZH_L1_PARSE=parallel timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slots', $sl, d['kernels_ms'])"; done
