# round 6, pass i: the exact matcher's match records gathered in LDS and written 64 at a time as whole 128-byte lines
# (ZH_L1_RECBUF=1) instead of a partial write a batch: are the 455 M partial writes a GiB what the kernel waits for?
export TMPDIR=/tmp
O=gpurun_out
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run ""; run _recbuf; done
echo "== one GPU's share"
for rep in 1 2; do run "" --buffers 512; run _recbuf --buffers 512; done
ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_recbuf.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or level1 or identical_all_levels or tokens or reruns" 2>&1 | tail -2
ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_recbuf.so bash tools/prof/pmc_sq.sh r06_i_recbuf 1024 l1 > $O/r06_i_pmc.log 2>&1
grep -A14 "zh_l1_match_kernel" $O/r06_i_recbuf_pmc_sq.txt | head -16
grep -E "TCC_EA0_WRREQ" $O/r06_i_recbuf_pmc_sq.txt | head -3
