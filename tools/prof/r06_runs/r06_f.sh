# round 6, pass f: bench.py's line with the new configs (unsized streams; one GPU's share over its own PCIe link: four legs
# and the pipelined trip) and the summary as its last key; the GPU suite on the current sources.
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/r06_f_bench.err | tail -1 > $O/r06_f_bench.json
tail -5 $O/r06_f_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_f_bench.json"))
print(json.dumps(d["summary"]))
print(json.dumps(d["configs"]["share512"].get("own_pcie_link")))
print(json.dumps(d["configs"].get("unsized")))
print(len(json.dumps(d)))
PY
