# round 6, pass p: contract mode's matcher with a thread's 32 table answers in registers before its walks
# (ZH_L1P_PRELOAD=1: no load inside a walk; 64 VGPRs, 16 spilled) against the product build; sizes must not change.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so ZH_L1_PARSE=parallel timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --no-parity-sample --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d['ratio'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run ""; run _l1ppre; done
echo "== one GPU's share"
for rep in 1 2; do run "" --buffers 512; run _l1ppre --buffers 512; done
