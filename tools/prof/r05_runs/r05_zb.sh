# round 5, pass zb: uncompress batches as two halves on two streams (ZH_INFLATE_HALVES=0: one launch a kernel)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fixtures or identical or damaged or config3 or ragged or plan or foreign" 2>&1 | tail -2
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('value_parallel_parse'), d.get('uncompress_GiBps'), d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms'].items() if 'inflate' in k or 'waiting' in k}, d.get('kernel_launches'))"; }
echo "== full"; for r in 1 2; do run ZH_INFLATE_HALVES=2048; run ZH_INFLATE_HALVES=0; done
echo "== zlib6"; run ZH_INFLATE_HALVES=2048 --foreign 6; run ZH_INFLATE_HALVES=0 --foreign 6
