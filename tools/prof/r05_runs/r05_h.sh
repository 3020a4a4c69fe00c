export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
run() { env $1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$2.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:3}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 lib$2', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d['uncompress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
echo "== c4 share"; for rep in 1 2; do for v in "" _chunk16 _chunk64 _lt2 _lt5; do run X=1 "$v" --buffers 512 --level -1 --compress-only --no-parallel-parse; done; done 2>&1 | tee $O/r05_h_c4_variants.txt
echo "== full"; run X=1 "" --no-parallel-parse
