# round 6, pass c: the writer's end of a round with the fence in front of the ring's fetch (pass b had it behind: every
# other round waited for the loads it had just issued), and the sizing pass of streams without a size on the tokens kernel
# (its count-only instantiation) instead of the serial decoder.  A/B against the round-5 library on ONE box.
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('value_parallel_parse'), d.get('compress_GiBps'), d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do for v in _r05 ""; do run "$v" --uncompress-only --no-parallel-parse; done; done
echo "== uncompress only, zlib-6 members"
for rep in 1 2; do for v in _r05 ""; do run "$v" --foreign 6 --no-parallel-parse; done; done
echo "== headline"
for v in _r05 ""; do run "$v"; done
echo "== streams without a size"
for v in _r05 ""; do ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$v.so timeout 600 python tools/bench_unsized.py 2>&1 | tail -1; done
ZH_TRACE=1 timeout 600 python tools/bench_unsized.py --buffers 1024 --reps 1 2>&1 | grep -v "^{" | tail -12
