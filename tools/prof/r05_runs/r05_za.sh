# round 5, pass za (experiment): the uncompress batch in parts on two streams (a part's writer beside the next part's tokens kernel)
export TMPDIR=/tmp
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('uncompress_GiBps'), d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms'].items() if 'inflate' in k or 'checksum' in k})"; }
echo "== uncompress only"; for h in 0 2 4 8 0; do run ZH_INFLATE_HALVES=$h --uncompress-only; done
echo "== zlib6"; for h in 0 2 4; do run ZH_INFLATE_HALVES=$h --foreign 6; done
