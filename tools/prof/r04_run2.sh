O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_b_pytest_gpu.log 2>&1; tail -3 $O/r04_b_pytest_gpu.log
for v in "" _nockpt _serhdr; do
  L=$(pwd)/zippy_amd/libzippy_hip$v.so
  echo "== lib $v"
  ZIPPY_HIP_LIB=$L timeout 300 python bench.py --uncompress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('own', d['value'], d['kernels_ms'])"
  ZIPPY_HIP_LIB=$L timeout 300 python bench.py --foreign 6 --steps 5 --warmup 1 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('zlib6', d['value'], d['kernels_ms'])"
done
timeout 300 python tools/kprof.py --foreign 6 --buffers 1024 > $O/r04_b_kprof_foreign6.txt 2>&1; grep -A10 "kernel ms\|tokens_kernel" $O/r04_b_kprof_foreign6.txt | head -24
timeout 300 python tools/kprof.py --buffers 1024 > $O/r04_b_kprof_own.txt 2>&1; grep -A10 "tokens_kernel" $O/r04_b_kprof_own.txt | head -12
