#!/bin/bash
# A SHORT evidence pass for a late change (the GPU budget's last minutes): the GPU tests, a quick bench line that the
# change has to show in (the parallel matcher under $1 ms, else nothing further is spent), then the PMC passes -- the
# bench lines quote hbm_traffic.json only for the sources it was measured on -- and, if time is left, the full line.
#   bash tools/prof/r04_p_pass.sh 22.3 r04_p
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; T=${2:-r04_p}
timeout 200 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.log 2>&1; tail -1 $O/${T}_pytest_gpu.log
grep -q " passed" $O/${T}_pytest_gpu.log && ! grep -q "failed" $O/${T}_pytest_gpu.log || { echo "GPU TESTS FAILED"; exit 2; }
timeout 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/${T}_quick.json
python - "$1" $O/${T}_quick.json <<'PY' || exit 3
import json, sys
d = json.load(open(sys.argv[2])); pp = d["parallel_parse"]
ms = pp["kernels_ms"]["zh_l1p_match_kernel"]
print("value", d["value"], "value_parallel_parse", d["value_parallel_parse"], "zh_l1p_match_kernel", round(ms, 2), "ms", "size_vs_exact", pp.get("size_vs_exact_parse"))
sys.exit(0 if ms < float(sys.argv[1]) else 3)
PY
bash tools/prof/pmc_passes.sh > $O/${T}_pmc.log 2>&1
cp $O/hbm_traffic.json profiles/hbm_traffic.json
timeout 300 python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/${T}_bench.json
cut -c1-300 $O/${T}_bench.json
