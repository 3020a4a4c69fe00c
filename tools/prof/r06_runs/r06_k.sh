# round 6, pass k: the exact matcher's wave count against the ROUNDS it makes: one GPU's share is 16 384 fragments -- 3.37
# rounds of 4864 waves, 3.0 of 5462 (22 waves a CU fit the LDS) -- and the full batch 131 072.
export TMPDIR=/tmp
run() { timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --no-parity-sample --compress-only "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do for sl in 4864 5120 5376 5462 5504 5632; do echo "share, slots $sl"; ZH_L1_SLOTS=$sl run --buffers 512; done; done
for sl in 4864 5462 5632; do echo "full batch, slots $sl"; ZH_L1_SLOTS=$sl run; done
for sl in 4864 5462; do echo "1024 buffers, slots $sl"; ZH_L1_SLOTS=$sl run --buffers 1024; done
