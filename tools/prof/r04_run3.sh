O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_c_pytest_gpu.log 2>&1; tail -3 $O/r04_c_pytest_gpu.log
timeout 300 python tools/kprof.py --foreign 6 --buffers 1024 --lib $(pwd)/zippy_amd/libzippy_hip_kprof_hdr.so > $O/r04_c_kprof_foreign6_hdr.txt 2>&1; grep -A10 "kernel ms\|tokens_kernel" $O/r04_c_kprof_foreign6_hdr.txt | head -24
timeout 300 python tools/kprof.py --l1-parse 1 --buffers 1024 > $O/r04_c_kprof_contract.txt 2>&1; grep -A10 "kernel ms\|== zh_huffman_kernel" $O/r04_c_kprof_contract.txt | head -30
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>$O/r04_c_bench.err | tail -1 > $O/r04_c_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_c_bench.json'))
print('value', d['value'], 'pp', d['value_parallel_parse'], d['parallel_parse']['size_vs_exact_parse'], d['kernels_ms'])
print('pp kernels', d['parallel_parse']['kernels_ms'])
for k,v in d['configs'].items(): print(k, v['value'], v['ms_per_step'], v['ratio'], v['dominant_kernel'], v['dominant_kernel_ms'])
PY
timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1; ZH_L1_PARSE=parallel timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1
