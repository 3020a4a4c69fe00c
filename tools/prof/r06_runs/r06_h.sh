# round 6, pass h: the exact matcher with a SECOND tag bit a slot, kept in LDS and looked at before the table is read
# (ZH_L1_LDSTAG=1: a probe whose bit differs skips the 128-byte read of a 2-byte entry that could only say "no"; 2 KiB more
# LDS a wave: 17 waves a CU instead of 19), against the product build on ONE box; then its read requests counted.
export TMPDIR=/tmp
O=gpurun_out
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do
  run ""; run _ldstag
  for sl in 4352 3840 5120; do echo "slots $sl"; ZH_L1_SLOTS=$sl run _ldstag; done
done
ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_ldstag.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or level1 or identical_all_levels or tokens or reruns" 2>&1 | tail -2
ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_ldstag.so bash tools/prof/pmc_sq.sh r06_h_ldstag 1024 l1 > $O/r06_h_pmc.log 2>&1
grep -A14 "zh_l1_match_kernel" $O/r06_h_ldstag_pmc_sq.txt | head -16
