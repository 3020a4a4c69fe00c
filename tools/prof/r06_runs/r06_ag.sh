# round 6, pass ag: the uncompress batch's two halves at other shares than 50 : 50 (ZH_INFLATE_SPLIT_PCT = the first half's
# share of the streams; the first half's checksum runs beside the second half's tail, the second half's behind the join)
export TMPDIR=/tmp
run() { ZH_INFLATE_SPLIT_PCT=$1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_split.so timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --no-parity-sample --uncompress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('first half $1 %', d['value'], d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.5})"; }
for rep in 1 2; do for p in 50 53 56 60 45; do run $p; done; done
echo "== zlib level-6 members"
for p in 50 53 56 60; do run $p --foreign 6; done
