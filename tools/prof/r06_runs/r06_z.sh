# round 6, pass z: what the driver runs at round end, on the round's last tree -- the GPU suite (52 tests with the pool-memory
# switch), smoke(), and `python bench.py` with no flags (its wall time next to the line).
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
timeout 1200 python bench.py 2>$O/r06_z_bench_no_flags.err | tail -1 > $O/r06_z_bench_no_flags.json
t1=$(date +%s)
echo "bench.py (no flags): $((t1 - t0)) s wall"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_z_bench_no_flags.json"))
print(json.dumps(d["summary"]))
print("traffic", d["roofline"]["traffic"], "line bytes", len(json.dumps(d)))
PY
