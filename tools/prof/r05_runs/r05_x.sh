# round 5, pass x: the checksum with 16 / 32 / 64 contiguous bytes a lane and row (the skip's four look-ups once per 16 / 32 / 64 bytes)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('value_parallel_parse'), {k:round(v,3) for k,v in d['kernels_ms'].items() if 'checksum' in k or 'huffman' in k}, d.get('kernel_launches'))"; }
echo "== uncompress only (one checksum pass of 4 GiB)"; for v in "" _ck32 _ck64; do run "$v" --uncompress-only; done
echo "== full"; for v in "" _ck32 _ck64; do run "$v"; done
ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_ck64.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "identical or fixtures or ragged or zip or config" 2>&1 | tail -2
