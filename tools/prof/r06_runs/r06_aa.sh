# round 6, pass aa: the chain walk at fewer workgroups a CU (unused dynamic LDS asked for at launch: ZH_WALK_LDS_KB), to
# see whether a smaller footprint an XCD -- 256 workgroups x 8192 positions = 8 MiB of records against 4 MiB of L2 -- buys
# more than the lanes it costs.  DefaultCompression on one GPU's share, parity sample on.
export TMPDIR=/tmp
run() { ZH_WALK_LDS_KB=$1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_wpad.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad KiB $1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for p in 0 24 32 40 53 64 0; do run $p; done
echo "== level 9 (128 buffers)"
for p in 0 32 53; do run $p --level 9 --buffers 128; done
