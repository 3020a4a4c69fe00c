# round 5, pass l: the Huffman replay with three heap levels a trip (and the wave around it), emit at 8 positions a lane
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "huffman or identical or fixtures or chain_levels or parallel_parse" > $O/r05_l_pytest_sub.log 2>&1; tail -2 $O/r05_l_pytest_sub.log
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.15}, (d.get('parallel_parse') or {}).get('kernels_ms',{}).get('zh_emit_kernel'))"; }
echo "== full"; for rep in 1 2; do run ""; run _emit4; done
echo "== share512"; run "" --buffers 512; run _emit4 --buffers 512
echo "== c2"; run "" --buffers 1024 --size 65536 --compress-only --steps 20; run _emit4 --buffers 1024 --size 65536 --compress-only --steps 20
echo "== kprof exact"; timeout 300 python tools/kprof.py --buffers 1024 > $O/r05_l_kprof_exact.txt 2>&1; grep -A10 "== zh_huffman_kernel" $O/r05_l_kprof_exact.txt; grep -A10 "== zh_emit_kernel" $O/r05_l_kprof_exact.txt
echo "== 1 GiB of noise, one stream"; timeout 300 python tools/gpu_big_buffer.py --mib 1024 --kind rand --no-oracle --no-zlib 2>&1 | tail -1
