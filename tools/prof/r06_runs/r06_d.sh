# round 6, pass d: which part of the writer's new round costs what pass b / c measured (15.8 -> 16.3 ms)?  Builds of the
# same sources: wrold = both parts off (round 5's round), wrcut0 = the one-barrier end only, wrend0 = the cut / carries by
# their owners only, "" = both; the round-5 library as the reference.  Uncompress only, one box.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do for v in _r05 _wrold _wrcut0 _wrend0 ""; do run "$v" --uncompress-only --no-parallel-parse; done; done
echo "== zlib-6 members"
for v in _r05 _wrold _wrcut0 _wrend0 ""; do run "$v" --foreign 6 --no-parallel-parse; done
echo "== without the halves (one launch each)"
for v in _r05 _wrold _wrcut0 _wrend0 ""; do ZH_INFLATE_HALVES=0 run "$v" --uncompress-only --no-parallel-parse; done
