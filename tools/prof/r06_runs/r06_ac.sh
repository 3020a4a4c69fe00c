# round 6, pass ac: contract mode's matcher (zh_l1p_match_kernel) leaves the coverage bitmap as well; the GPU suite on these sources
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); pp=d.get('parallel_parse') or {}; print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 1.0}, '| contract', d.get('value_parallel_parse'), (pp.get('parity_sample') or {}), {k:round(v,3) for k,v in (pp.get('kernels_ms') or {}).items() if v > 1.0})"; }
for rep in 1 2 3; do run _base; run ""; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
