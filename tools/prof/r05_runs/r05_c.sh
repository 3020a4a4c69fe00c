export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
run() { env $1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$2.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:3}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 lib$2', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d['uncompress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
echo "== full"; for rep in 1 2; do run ZH_L1_ORDER=0 "" --no-parallel-parse; run ZH_L1_TAIL_ROUNDS=1 "" --no-parallel-parse; run ZH_L1_TAIL_ROUNDS=2 "" --no-parallel-parse; run ZH_L1_TAIL_ROUNDS=4 "" --no-parallel-parse; run X=1 _occ5 --no-parallel-parse; done 2>&1 | tee $O/r05_c_ab.txt
echo "== share512"; for rep in 1 2; do run ZH_L1_ORDER=0 "" --buffers 512 --no-parallel-parse; run ZH_L1_TAIL_ROUNDS=1 "" --buffers 512 --no-parallel-parse;  run ZH_L1_TAIL_ROUNDS=2 "" --buffers 512 --no-parallel-parse; done 2>&1 | tee $O/r05_c_ab_share.txt
echo "== zlib6"; for rep in 1 2; do run X=1 "" --foreign 6; run X=1 _occ5 --foreign 6; done 2>&1 | tee $O/r05_c_ab_zlib6.txt
