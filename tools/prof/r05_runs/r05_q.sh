# round 5, pass q: the exact matcher's statistics phase (histograms + coverage bitmap + literal histogram) with its loads ahead
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "identical or fixtures or config2 or ragged or reruns" > $O/r05_q_pytest_sub.log 2>&1; tail -2 $O/r05_q_pytest_sub.log
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample --no-parallel-parse "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], d['compress_GiBps'], d.get('uncompress_GiBps'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.5})"; }
echo "== full"; for r in 1 2; do run ""; run _statsr4; done
echo "== share512"; run "" --buffers 512; run _statsr4 --buffers 512
echo "== c2"; run "" --buffers 1024 --size 65536 --compress-only --steps 20; run _statsr4 --buffers 1024 --size 65536 --compress-only --steps 20
