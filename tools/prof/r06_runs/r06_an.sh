# round 6, pass an: the chain links with 4 / 8 classes of a block as the waves of ONE workgroup that crosses the block tile by
# tile (a barrier every 8192 positions), so that what the classes write into a line of the link array arrives while the L2
# still holds the line (now: a wave a class, each at its own pace; 92 % of the 4-byte stores go on to the fabric alone).
# DefaultCompression on one GPU's share, parity sample on.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.5})"; }
for rep in 1 2; do run ""; run _lg4; run _lg8; done
echo "== level 9, level 3 (128 buffers)"
for l in 9 3; do run "" --level $l --buffers 128; run _lg8 --level $l --buffers 128; done
