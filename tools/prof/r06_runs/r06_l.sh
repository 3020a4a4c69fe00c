# round 6, pass l: the exact matcher's step with what only the hashes decide -- the slot counters, the shared-slot lanes (Cw) --
# worked out WHILE the table's answer is on its way instead of behind it (build earlycw) against the product build.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run ""; run _earlycw; done
echo "== one GPU's share"
for rep in 1 2; do run "" --buffers 512; run _earlycw --buffers 512; done
echo "== config 2"
for rep in 1 2; do run "" --buffers 1024 --size 65536; run _earlycw --buffers 1024 --size 65536; done
