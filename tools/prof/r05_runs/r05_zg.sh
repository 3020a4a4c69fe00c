# round 5, pass zg: the exact matcher on 4608 / 4864 / 5120 waves, three rounds; and one GPU's share
export TMPDIR=/tmp
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], {k:round(v,3) for k,v in d['kernels_ms'].items() if 'l1_match' in k})"; }
for r in 1 2 3; do for sl in 4608 4864 5120; do run ZH_L1_SLOTS=$sl; done; done
echo "== 512"; for sl in 4608 4864 5120 4608 5120; do run ZH_L1_SLOTS=$sl --buffers 512 --steps 10; done
