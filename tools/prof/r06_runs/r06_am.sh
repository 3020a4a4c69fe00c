# round 6, pass am: the chain walk's finished-search store issued behind the turn's loads have come back (ZH_CHAIN_LATE_STORE=1)
# instead of in front of them: a wait for a load also waits for every store before it (one counter), and a store's round trip
# is the longer one (551 against 399 cycles, r06_ak).  DefaultCompression on one GPU's share, parity sample on.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.5})"; }
for rep in 1 2 3; do run ""; run _late; done
echo "== level 9, level 3 (128 buffers)"
for l in 9 3; do run "" --level $l --buffers 128; run _late --level $l --buffers 128; done
