# round 5, pass zd (experiment): the two halves unequal -- the first stream's share in percent (it starts second and ends first)
export TMPDIR=/tmp
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('uncompress_GiBps'), d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms'].items() if 'waiting' in k})"; }
echo "== own"; for t in 50 56 62 68; do run ZH_EXP_SPLIT=$t --uncompress-only; done
echo "== zlib6"; for t in 50 56 62; do run ZH_EXP_SPLIT=$t --foreign 6; done
