O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/r04_a_pytest_gpu.log 2>&1; tail -3 $O/r04_a_pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 2>$O/r04_a_bench.err | tail -1 > $O/r04_a_bench.json; cut -c1-1500 $O/r04_a_bench.json; tail -3 $O/r04_a_bench.err
timeout 300 python tools/kprof.py --foreign 6 --buffers 1024 > $O/r04_a_kprof_foreign6.txt 2>&1; grep -A12 "kernel ms\|tokens_kernel" $O/r04_a_kprof_foreign6.txt | head -40
