# round 5, pass t: chained stored blocks as segment starts (one large stream of incompressible data on many workgroups)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seg or stored or one_gib or stream or tar or zip or fixtures" > $O/r05_t_pytest_sub.log 2>&1; tail -3 $O/r05_t_pytest_sub.log
for k in "--kind rand" "--level -2" "--kind text" "--level 6"; do timeout 200 python tools/gpu_big_buffer.py --mib 1024 --no-oracle --no-zlib $k 2>/dev/null | tail -1 | cut -c1-900; done
timeout 600 python tools/bench_one_stream.py 2>/dev/null | tail -1
timeout 900 python tools/gpu_fuzz.py --seg-mutations 2000 2>&1 | tail -2
