O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "parallel or smoke or lds" > $O/r04_f_pytest_gpu.log 2>&1; tail -2 $O/r04_f_pytest_gpu.log
timeout 300 python tools/kprof.py --l1-parse 1 --buffers 1024 > $O/r04_f_kprof_contract.txt 2>&1; grep -A10 "kernel ms\|== zh_l1p_match" $O/r04_f_kprof_contract.txt | head -22
for sl in 512 768; do ZH_L1P_SLOTS=$sl ZH_L1_PARSE=parallel timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slots', $sl, d['value'], d['kernels_ms'])"; tail -2 $O/err.txt; done
