# round 6, pass ap: the chain walk's best[] / compare reads as non-temporal loads (ZH_CHAIN_NT=1), to leave the L2 to the link
# records.  DefaultCompression on one GPU's share, parity sample on.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.5})"; }
for rep in 1 2 3; do run ""; run _nt1; done
echo "== level 9 (128 buffers)"
run "" --level 9 --buffers 128; run _nt1 --level 9 --buffers 128
