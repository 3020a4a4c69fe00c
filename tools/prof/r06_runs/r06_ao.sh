# round 6, pass ao: the grouped chain links (r06_an) at other tile sizes and with 64 classes (16 or 8 of them a workgroup)
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.5})"; }
for rep in 1 2 3; do for v in "" _lg8 _lg8t16 _lg8t32 _lg8t64; do run "$v"; done; done
