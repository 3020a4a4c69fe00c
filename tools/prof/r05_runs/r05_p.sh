# round 5, pass p: checksum through 11 + 11 + 10-bit tables (3 lookups a dword), emit's cover bitmap from flips + prefix parity
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "huffman or identical or fixtures or chain_levels or parallel_parse or ragged or zip" > $O/r05_p_pytest_sub.log 2>&1; tail -2 $O/r05_p_pytest_sub.log
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.15}, d.get('kernel_launches'))"; }
echo "== full"; run ""; run ""
echo "== share512"; run "" --buffers 512
echo "== c2"; run "" --buffers 1024 --size 65536 --compress-only --steps 20
echo "== c3 own"; run "" --uncompress-only
