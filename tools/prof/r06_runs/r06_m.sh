# round 6, pass m: emission with a pass's eight literal codes looked up together, unconditionally (ZH_EMIT_HOIST=1), instead
# of under each position's own condition (the compiler waits for each LDS look-up where it is issued).
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run ""; run _ehoist; done
echo "== one GPU's share"
for rep in 1 2; do run "" --buffers 512; run _ehoist --buffers 512; done
