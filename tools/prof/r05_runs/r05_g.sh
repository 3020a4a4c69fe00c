export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stored or chain or config4 or levels or fixtures or tokens" > $O/r05_g_pytest_gpu.log 2>&1; tail -3 $O/r05_g_pytest_gpu.log
run() { env $1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$2.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:3}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 lib$2', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d['uncompress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
echo "== full"; run X=1 "" --no-parallel-parse; run X=1 "" --no-parallel-parse
echo "== share512"; run X=1 "" --buffers 512 --no-parallel-parse
echo "== 256 streams"; run X=1 "" --buffers 256 --no-parallel-parse
echo "== c4 share"; run X=1 "" --buffers 512 --level -1 --compress-only --no-parallel-parse; run X=1 "" --buffers 512 --level -1 --compress-only --no-parallel-parse
echo "== c4 level 9 / 5 / 3"; run X=1 "" --buffers 128 --level 9 --compress-only --no-parallel-parse; run X=1 "" --buffers 256 --level 5 --compress-only --no-parallel-parse; run X=1 "" --buffers 256 --level 3 --compress-only --no-parallel-parse
(for k in "--kind rand"; do timeout 300 python tools/gpu_big_buffer.py --mib 1024 --no-oracle --no-zlib $k 2>/dev/null | tail -1; done) > $O/r05_g_big_buffer.json; cut -c1-900 $O/r05_g_big_buffer.json
timeout 600 python tools/gpu_fuzz_chain.py 700 20 2>&1 | tail -2
