# round 6, pass q: the chain walk's turn with its record load as two unconditional four-byte loads (ZH_CHAIN_SPLIT_LOAD=1)
# instead of a four-byte load under one condition and an eight-byte one under its complement -- which the compiler had
# turned into one trip to memory behind the other.  DefaultCompression on one GPU's share, parity sample on.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run ""; run _csplit; done
echo "== level 9, level 3 (wide records)"
for l in 9 3; do run "" --level $l --buffers 128; run _csplit --level $l --buffers 128; done
