# round 5, pass zc (experiment): the uncompress batch in three parts on two streams against two halves
export TMPDIR=/tmp
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('uncompress_GiBps'), d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms'].items() if 'waiting' in k})"; }
echo "== own"; for t in 0 1 2 0; do run ZH_EXP_THIRD=$t --uncompress-only; done
echo "== zlib6"; for t in 0 1 2; do run ZH_EXP_THIRD=$t --foreign 6; done
