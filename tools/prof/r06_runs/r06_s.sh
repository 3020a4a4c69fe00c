# round 6, pass s: emission's match-bitmap chunk: 2 / 4 / 8 KiB of positions a build (LDS 5.9 / 6.4 / 7.4 KiB a wave)
# -- the generalised parity code (ec4k) against the product build first.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do for v in "" _ec4k _ec8k _ec2k; do run "$v"; done; done
