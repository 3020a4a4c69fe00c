# round 5, pass zh: the checksum's epilogue without gf2_xpow8() for whole pieces
export TMPDIR=/tmp
run() { timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if 'checksum' in k})"; }
run --uncompress-only; run --uncompress-only
