# round 5, pass n: the Huffman replay with the wave doing a pop (a node of the five levels below the hole a lane)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "huffman or identical or fixtures or chain_levels or parallel_parse" > $O/r05_n_pytest_sub.log 2>&1; tail -2 $O/r05_n_pytest_sub.log
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.15})"; }
echo "== full"; run ""
echo "== share512"; run "" --buffers 512
echo "== c2"; run "" --buffers 1024 --size 65536 --compress-only --steps 20
echo "== kprof exact"; timeout 300 python tools/kprof.py --buffers 1024 > $O/r05_n_kprof_exact.txt 2>&1; grep -A10 "== zh_huffman_kernel" $O/r05_n_kprof_exact.txt | head -12
