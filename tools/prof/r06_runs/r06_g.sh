# round 6, pass g: the first half's checksum pieces in front of the wait for the second half (uncompress batches that run as
# two halves), against the round-5 library; the PCIe trip of one GPU's share at 1 / 2 / 4 chunks.
export TMPDIR=/tmp
O=gpurun_out
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do for v in _r05 ""; do run "$v" --uncompress-only --no-parallel-parse; done; done
echo "== zlib-6 members"
for rep in 1 2; do for v in _r05 ""; do run "$v" --foreign 6 --no-parallel-parse; done; done
timeout 600 python -m pytest tests -m gpu -x -q -k "fixtures or damaged or unsized or plan" 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parallel-parse 2>/dev/null | tail -1 > $O/r06_g_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_g_bench.json"))
print(json.dumps(d["summary"]))
print(json.dumps(d["configs"]["share512"].get("own_pcie_link")))
PY
