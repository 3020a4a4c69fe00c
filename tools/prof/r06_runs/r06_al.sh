# round 6, pass al: the parallel matcher's link phase with loads only in wave 0's vector-memory queue: its results go to an
# LDS ring and the waiting waves carry them out (before: the wave's own stores shared the counter with its prefetched loads,
# and the compiler waited for everything -- s_waitcnt vmcnt(0) -- at the head of every block of 16 steps).
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); pp=d.get('parallel_parse') or {}; print('lib$1', d['value'], '| contract', d.get('value_parallel_parse'), pp.get('ms_per_step'), pp.get('size_vs_exact_parse'), (pp.get('parity_sample') or {}), {k:round(v,3) for k,v in (pp.get('kernels_ms') or {}).items() if v > 4.0})"; }
for rep in 1 2 3; do run ""; run _p1ring; done
echo "== one GPU's share; config 2"
for rep in 1 2; do run "" --buffers 512; run _p1ring --buffers 512; done
run "" --buffers 1024 --size 65536; run _p1ring --buffers 1024 --size 65536
