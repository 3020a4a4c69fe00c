# round 6, pass x: the inflate pair's tunables on round 5's decode loop (they were set in rounds 2-4): the run-up 128 / 256 /
# 384 / 512 bits, subchunks of 1024 bits, the writer at 4 / 5 / 6 workgroups a CU.  Uncompress only, own streams and
# zlib-6 members.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do for v in "" _ru128 _ru256 _ru384 _sb1024 _wr4 _wr6; do run "$v" --uncompress-only; done; done
echo "== zlib-6 members"
for v in "" _ru128 _ru256 _ru384 _sb1024 _wr4 _wr6; do run "$v" --foreign 6; done
echo "== one launch each (no halves)"
for v in "" _ru256 _ru384 _wr4 _wr6; do ZH_INFLATE_HALVES=0 run "$v" --uncompress-only; done
