# A/B of builds of the library on the same box: bash tools/prof/r05_ab.sh "<suffix> <suffix> ..." [bench args]
# ("" = the product library; boxes differ by ~ 7 %: never compare across runs)
V=$1; shift
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('value_parallel_parse'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do for v in $V; do run "${v#-}" "$@"; done; done
