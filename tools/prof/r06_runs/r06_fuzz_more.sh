# round 6: more fuzz on the final sources than the final pass takes -- fresh seed ranges
export TMPDIR=/tmp
(timeout 2400 python tools/gpu_fuzz.py 4000 300 2>&1 | tail -3
 timeout 1500 python tools/gpu_fuzz_chain.py 1200 80 2>&1 | tail -2
 timeout 1800 python tools/gpu_fuzz.py --mutations 30000 2>&1 | tail -3
 timeout 1200 python tools/gpu_fuzz.py --seg-mutations 8000 2>&1 | tail -2) > gpurun_out/r06_z_fuzz_more.log 2>&1
cat gpurun_out/r06_z_fuzz_more.log
