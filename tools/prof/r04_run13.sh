O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_i_pytest_gpu.log 2>&1; tail -2 $O/r04_i_pytest_gpu.log
timeout 900 python tools/gpu_fuzz.py 4000 30 2>&1 | tail -3
timeout 900 python tools/gpu_fuzz_chain.py 700 6 2>&1 | tail -2
(timeout 900 python tools/gpu_fuzz.py --mutations 4000 2>&1 | tail -2; timeout 900 python tools/gpu_fuzz.py --seg-mutations 1000 2>&1 | tail -2)
