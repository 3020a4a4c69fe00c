# round 6, pass r: the uncompress batch's two halves STAGGERED (ZH_INFLATE_STAGGER=1: the second half's tokens kernel starts
# when the first half's has ended, beside the first half's writer) against both halves starting together.
export TMPDIR=/tmp
run() { timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do echo together; run --uncompress-only; echo staggered; ZH_INFLATE_STAGGER=1 run --uncompress-only; done
echo "== zlib-6 members"
for rep in 1 2; do echo together; run --foreign 6; echo staggered; ZH_INFLATE_STAGGER=1 run --foreign 6; done
