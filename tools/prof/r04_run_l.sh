O=gpurun_out; mkdir -p $O
bash tools/prof/pmc_passes.sh > $O/r04_l_pmc.log 2>&1; cat $O/r04_l_pmc.log
mkdir -p profiles; cp $O/hbm_traffic.json profiles/hbm_traffic.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_l_pytest_gpu.log 2>&1; tail -2 $O/r04_l_pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 2>$O/r04_l_bench.err | tail -1 > $O/r04_l_bench.json
timeout 300 python tools/kprof.py --foreign 6 --buffers 1024 > $O/r04_l_kprof_foreign6.txt 2>&1
cd /tmp; rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $OLDPWD/$O/r04_l_rocprof_bench.log 2>&1; cd $OLDPWD
python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/r04_l_kernel_stats.csv 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_l_bench.json'))
print('value', d['value'], d['value_parallel_parse'], d['roofline']['traffic'])
for k,v in d['configs'].items(): print(k, v['value'], v['dominant_kernel'], v['dominant_kernel_ms'], v['frac'], v['traffic'])
PY
head -8 $O/r04_l_kernel_stats.csv | cut -c1-200
