export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05_i_pytest_gpu.log 2>&1; tail -2 $O/r05_i_pytest_gpu.log
run() { env $1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$2.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:3}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 lib$2', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d['uncompress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
echo "== full"; for rep in 1 2; do run X=1 ""; run ZH_CHECKSUM_ASIDE=0 ""; done
echo "== share512"; run X=1 "" --buffers 512; run ZH_CHECKSUM_ASIDE=0 "" --buffers 512
echo "== c2"; run X=1 "" --buffers 1024 --size 65536 --compress-only; run ZH_CHECKSUM_ASIDE=0 "" --buffers 1024 --size 65536 --compress-only
