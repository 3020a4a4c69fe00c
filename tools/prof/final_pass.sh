#!/bin/bash
# The per-round evidence under profiles/ (run on the GPU box from the repo root):
#   bash tools/prof/final_pass.sh r01_g
# -> gpurun_out/<tag>_bench.json, _kernel_stats.csv, _pytest_gpu.log, _c5.json, _host_api.json, hbm_traffic.json
R=$(pwd); T=${1:-r01}
O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > $O/${T}_pytest_gpu.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 1 2>/dev/null | tail -1 > $O/${T}_bench.json
timeout 300 python tools/bench_c5.py --plain 2>/dev/null | tail -1 > $O/${T}_c5.json
timeout 300 python tools/bench_c5.py --level -1 --steps 2 2>/dev/null | tail -1 >> $O/${T}_c5.json
timeout 300 python tools/bench_host_api.py --reps 2 2>/dev/null | tail -1 > $O/${T}_host_api.json
cd /tmp
rm -rf /tmp/kt /tmp/pf /tmp/pw
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${T}_rocprof_bench.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${T}_kernel_stats.csv 2>$O/${T}_summary.err
python tools/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) --buffers 4096 --size 1048576 > $O/hbm_traffic.json 2>>$O/${T}_summary.err
tail -2 $O/${T}_pytest_gpu.log; cat $O/${T}_bench.json | cut -c1-400; head -8 $O/${T}_kernel_stats.csv
