# round 6, pass t: emission's match-bitmap chunk, smaller: 2048 / 1024 / 512 positions a build (512 = a pass) against the
# product build's 4096; level -1 on a share too (long matches reach across many chunks).
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do for v in "" _ec2048 _ec1024 _ec512; do run "$v"; done; done
echo "== level -1, a share"
for v in "" _ec2048 _ec1024 _ec512; do run "$v" --buffers 512 --level -1; done
echo "== contract mode"
for v in "" _ec2048 _ec1024 _ec512; do ZH_L1_PARSE=parallel run "$v" --no-parity-sample; done
