# round 6, pass b: the compress run without its memset (the layout kernels zero exactly the words that are OR-ed into) and
# the writer's round with four barriers instead of seven, against the round-5 library (libzippy_hip_r05.so, built from
# 98781c3) on ONE box; then one GPU's share with other widths of the inflate pair; then streams without a size.
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('value_parallel_parse'), d.get('compress_GiBps'), d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do for v in _r05 ""; do run "$v"; done; done
echo "== uncompress only, zlib-6 members"
for rep in 1 2; do for v in _r05 ""; do run "$v" --foreign 6 --no-parallel-parse; done; done
echo "== one GPU's share (512 x 1 MiB)"
for v in _r05 ""; do run "$v" --buffers 512 --no-parallel-parse; done
echo "== the share, uncompress only: widths of the inflate pair (ZH_INFLATE_WIDE = largest batch on 1024 threads, MID: the writer's 512)"
for w in "768 640" "0 640" "0 0" "256 640" "256 0" "768 0" "768 768"; do set -- $w; echo "wide $1 mid $2"; ZH_INFLATE_WIDE=$1 ZH_INFLATE_MID=$2 run "" --buffers 512 --uncompress-only --no-parallel-parse; done
echo "== streams without a size"
timeout 600 python tools/bench_unsized.py 2>&1 | tail -1
ZH_TRACE=1 timeout 600 python tools/bench_unsized.py --buffers 256 --reps 1 2>&1 | grep -v "^{" | tail -40
