# round 6, pass n: emission -- literal codes looked up together (1), the lane's matches too (2), every position's two ORs
# without a branch (3) -- against the product build; contract mode and config 2 as well for the winner.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do for v in "" _ehoist _ehoist2 _ehoist3; do run "$v"; done; done
echo "== level -1, one GPU's share (long matches)"
for v in "" _ehoist _ehoist2 _ehoist3; do run "$v" --buffers 512 --level -1; done
