run() { ZH_WR_ONE=$1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$2.so timeout 300 python bench.py --uncompress-only --buffers $3 --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parity-sample 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one>=$1 lib$2 n=$3', d['value'], {k:v for k,v in d['kernels_ms'].items() if 'inflate' in k})"; }
run 0 "" 4096
run 1 "" 4096
run 1 _wr5 4096
run 0 "" 3072
run 1 "" 3072
