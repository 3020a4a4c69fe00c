run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 300 python bench.py --uncompress-only --buffers 4096 --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parity-sample 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], {k:v for k,v in d['kernels_ms'].items() if 'inflate' in k or 'checksum_pieces' in k})"; }
run ""
run _old
run ""
run _old
