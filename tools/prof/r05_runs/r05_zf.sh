# round 5, pass zf: the exact matcher's waves (ZH_L1_SLOTS: 20 a CU by default, LDS allows 21) on the last sources
export TMPDIR=/tmp
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse --compress-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], {k:round(v,3) for k,v in d['kernels_ms'].items() if 'l1_match' in k})"; }
for sl in 5120 5376 4608 4096 5120; do run ZH_L1_SLOTS=$sl; done
