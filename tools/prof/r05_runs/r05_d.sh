export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05_d_pytest_gpu.log 2>&1; tail -3 $O/r05_d_pytest_gpu.log
(for k in "--kind rand" "--level -2" "--kind text" "--level 6"; do timeout 300 python tools/gpu_big_buffer.py --mib 1024 --no-oracle --no-zlib $k 2>/dev/null | tail -1; done) > $O/r05_d_big_buffer.json; cut -c1-400 $O/r05_d_big_buffer.json
run() { env $1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$2.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:3}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 lib$2', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d['uncompress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
echo "== full"; run X=1 ""; run X=1 ""
echo "== share512"; run X=1 "" --buffers 512; run ZH_L1_ORDER=0 "" --buffers 512; run X=1 "" --buffers 512
