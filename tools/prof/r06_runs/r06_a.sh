# round 6, pass a: VERDICT r5 item 1 -- is the exact matcher's fabric limit BYTES or REQUESTS?  The table pool as an
# allocation of its own in uncached (hipDeviceMallocUncached) / fine-grained memory against the arena's ordinary memory
# (ZH_L1_POOL), A/B on ONE box: kernel time, then the request counters by size for each arm.
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse --compress-only "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do
  for m in default uncached fine; do echo "== pool $m"; ZH_L1_POOL=$m run; done
done
for sl in 3072 6144 7680; do echo "== uncached, $sl slots"; ZH_L1_POOL=uncached ZH_L1_SLOTS=$sl run; done
for sl in 3072 6144; do echo "== fine, $sl slots"; ZH_L1_POOL=fine ZH_L1_SLOTS=$sl run; done
echo "== parity with the pool uncached: the GPU identity tests"
ZH_L1_POOL=uncached timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or level1 or identical_all_levels" 2>&1 | tail -2
for m in default uncached fine; do
  ZH_L1_POOL=$m bash tools/prof/pmc_sq.sh r06_a_$m 1024 l1 > $O/r06_a_pmc_$m.log 2>&1
  grep -A12 "zh_l1_match_kernel" $O/r06_a_${m}_pmc_sq.txt | head -16
done
