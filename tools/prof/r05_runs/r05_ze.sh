# round 5, pass ze: two halves for one GPU's share (512 streams: the wide kernels) -- ZH_INFLATE_HALVES=512 against the default (2048: one launch)
export TMPDIR=/tmp
run() { env $1 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('uncompress_GiBps'), d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms'].items() if 'inflate' in k or 'waiting' in k})"; }
for b in 512 1024 1536; do echo "== $b streams"; run ZH_INFLATE_HALVES=$b --buffers $b; run ZH_INFLATE_HALVES=4096 --buffers $b; done
