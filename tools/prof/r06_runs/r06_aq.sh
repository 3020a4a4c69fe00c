# round 6, pass aq: the grouped chain links with 2 / 4 (product) / 8 / 16 steps of 64 positions a turn (their loads in flight
# together; every turn ends in one wait for all of them and the turn's stores)
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 2.5})"; }
for rep in 1 2 3; do for v in "" _la2 _la8 _la16; do run "$v"; done; done
