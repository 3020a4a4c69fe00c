# round 6, pass e: the chain walk with a WAVE's lanes taking the wave's chunks first come, first served (ZH_CHAIN_PER chunks a
# lane, ZH_CHAIN_CHUNK positions a chunk, ZH_CHAIN_OVER: how far a walk goes on behind its chunk), against one chunk a lane.
# One GPU's share at DefaultCompression (512 x 1 MiB), compress only; every variant's streams are checked by the bench's own
# round trip, the default's and the best one's against the oracle (parity sample on).
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do for v in "" _p2c16 _p2c16o16k _p2c16o256 _p4c8 _p2c32 _p4c16 _p4c32; do run "$v"; done; done
