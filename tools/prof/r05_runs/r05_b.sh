export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
(nproc; cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpuset.cpus.effective 2>&1; lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|NUMA node\(s\)|Model name") > $O/r05_b_host.txt 2>&1; cat $O/r05_b_host.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/r05_b_pytest_gpu.log 2>&1; tail -3 $O/r05_b_pytest_gpu.log
run() { env $1 ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$2.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:3}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 lib$2', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], d['uncompress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2; do
  run X=1 ""; run ZH_L1_ORDER=0 ""; run ZH_CHECKSUM_ASIDE=0 ""; run X=1 _base
done 2>&1 | tee $O/r05_b_ab.txt
echo "== share512"; for rep in 1 2; do run X=1 "" --buffers 512; run ZH_L1_ORDER=0 "" --buffers 512; run X=1 _base --buffers 512; done 2>&1 | tee $O/r05_b_ab_share.txt
timeout 900 python bench.py --steps 10 --warmup 2 2>$O/r05_b_bench.err | tail -1 > $O/r05_b_bench.json; cut -c1-1500 $O/r05_b_bench.json; tail -5 $O/r05_b_bench.err
