# round 6, pass o: emission, on top of the hoisted literal look-ups: the match that reaches into a chunk kept in registers
# from the chunk before (no dependent global loads by one lane at a chunk's head), the pass's bitmap words read
# unconditionally (build emit2) against the product build.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do for v in "" _emit2; do run "$v"; done; done
echo "== one GPU's share"
for rep in 1 2; do for v in "" _emit2; do run "$v" --buffers 512; done; done
