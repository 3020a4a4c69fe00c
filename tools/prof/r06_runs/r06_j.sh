# round 6, pass j: the exact matcher with the NEXT step's source bytes asked for before this step's walk (ZH_L1_WINDOW=1:
# 1 KiB a wave in registers, the lanes' 16 bytes cut out of it by ds_bpermute): the first of a step's three dependent
# memory trips off the chain.  A/B against the product build on ONE box.
export TMPDIR=/tmp
O=gpurun_out
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run ""; run _win; done
echo "== one GPU's share"
for rep in 1 2; do run "" --buffers 512; run _win --buffers 512; done
echo "== config 2 (1024 x 64 KiB)"
for rep in 1 2; do run "" --buffers 1024 --size 65536; run _win --buffers 1024 --size 65536; done
ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_win.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or level1 or identical_all_levels or tokens or reruns or edge" 2>&1 | tail -2
