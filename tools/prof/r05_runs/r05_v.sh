# round 5, pass v: the Huffman pop's first trip asked for with the item and the root, the leaves' frequencies a push ahead; the combine's piece table in LDS
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "huffman or identical or fixtures or config2" > $O/r05_v_pytest_sub.log 2>&1; tail -2 $O/r05_v_pytest_sub.log
timeout 300 python tools/kprof.py --buffers 1024 2>&1 | grep -A22 "== zh_huffman_kernel" | head -24
run() { timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "${@:1}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['compress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.1})"; }
echo "== share512"; run --buffers 512
echo "== c2"; run --buffers 1024 --size 65536 --compress-only --steps 20
timeout 200 python tools/gpu_big_buffer.py --mib 1024 --no-oracle --no-zlib --kind rand 2>/dev/null | tail -1 | cut -c1-700
