O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_d_pytest_gpu.log 2>&1; tail -2 $O/r04_d_pytest_gpu.log
timeout 300 python tools/kprof.py --foreign 6 --buffers 1024 --lib $(pwd)/zippy_amd/libzippy_hip_kprof_hdr.so > $O/r04_d_kprof_foreign6_hdr.txt 2>&1; grep -A10 "kernel ms\|tokens_kernel" $O/r04_d_kprof_foreign6_hdr.txt | grep -v "l1_match\|step:\|stage-in\|stats phase\|#steps\|#fast\|#events" | head -14
timeout 300 python tools/kprof.py --l1-parse 1 --buffers 1024 > $O/r04_d_kprof_contract.txt 2>&1; grep -A10 "== zh_huffman_kernel" $O/r04_d_kprof_contract.txt | head -11
timeout 300 python tools/kprof.py --buffers 1024 > $O/r04_d_kprof_exact.txt 2>&1; grep -A10 "== zh_huffman_kernel" $O/r04_d_kprof_exact.txt | head -11
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>$O/r04_d_bench.err | tail -1 > $O/r04_d_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_d_bench.json'))
print('value', d['value'], 'pp', d['value_parallel_parse'], d['parallel_parse']['size_vs_exact_parse'], d['kernels_ms'])
print('pp kernels', d['parallel_parse']['kernels_ms'])
for k,v in d['configs'].items(): print(k, v['value'], v['ms_per_step'], v['ratio'], v['dominant_kernel'], v['dominant_kernel_ms'])
PY
timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1; ZH_L1_PARSE=parallel timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1
