#!/bin/bash
# The per-round evidence under profiles/ (run on the GPU box from the repo root):
#   bash tools/prof/final_pass.sh r03_a
# -> gpurun_out/<tag>_bench.json (BASELINE metric + configs 2-5 + the parallel-parse leg + cpu_baseline),
#    _c4.json / _c4share.json (DefaultCompression, whole batch on one GPU / 512 x 1 MiB), _share512.json (one GPU's share of eight),
#    _kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench command), _pytest_gpu.log,
#    _host_api*.json, _single_call.json, _one_stream.json, _unsized.json, _big_buffer.json (ONE buffer of 4 GiB + 12345 bytes; 1 GiB streams), _fuzz*.log, hbm_traffic.json (tools/prof/pmc_passes.sh: two --pmc passes a workload)
R=$(pwd); T=${1:-r03}
O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
# the PMC passes first: profiles/hbm_traffic.json has to be there (with these sources' hash) for the bench lines to quote it
bash tools/prof/pmc_passes.sh > $O/${T}_pmc.log 2>&1   # -> gpurun_out/hbm_traffic.json: the headline and configs 2-5, a pair of PMC passes each
mkdir -p profiles && cp $O/hbm_traffic.json profiles/hbm_traffic.json
# scheduler / L2 counters per kernel (SQ_*, TCC_*: valu busy, lane utilisation, waits, L2 hit rate, requests by size)
bash tools/prof/pmc_sq.sh ${T} 1024 > $O/${T}_pmc_sq.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $O/${T}_pytest_gpu.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 2 2>$O/${T}_bench.err | tail -1 > $O/${T}_bench.json
timeout 600 python bench.py --level -1 --compress-only --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${T}_c4.json
timeout 300 python bench.py --buffers 512 --level -1 --compress-only --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse 2>/dev/null | tail -1 > $O/${T}_c4share.json
# one GPU's share of the batch when eight GPUs split it (strong scaling, 512 x 1 MiB)
timeout 300 python bench.py --buffers 512 --steps 10 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/${T}_share512.json
# the same two steps with the fragments as numbered in every run (ZH_L1_ORDER=0): what a plan's FIRST run costs, before
# the matcher knows what its fragments cost (the lines above are steady state: the plans have run before)
ZH_L1_ORDER=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --no-parity-sample 2>/dev/null | tail -1 > $O/${T}_bench_first_run_order.json
ZH_L1_ORDER=0 timeout 300 python bench.py --buffers 512 --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --no-parity-sample 2>/dev/null | tail -1 > $O/${T}_share512_first_run_order.json
timeout 300 python tools/bench_host_api.py --reps 2 2>/dev/null | tail -1 > $O/${T}_host_api.json
timeout 300 python tools/bench_host_api.py --reps 2 --buffers 4096 2>/dev/null | tail -1 > $O/${T}_host_api_4096.json
timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1 > $O/${T}_single_call.json
timeout 600 python tools/bench_one_stream.py 2>/dev/null | tail -1 > $O/${T}_one_stream.json
# streams without a size field (zlib / raw deflate that outgrow the 4 x guess) next to the same data as gzip members
timeout 600 python tools/bench_unsized.py 2>/dev/null | tail -1 > $O/${T}_unsized.json
# ONE buffer of more than 4 GiB against the oracle, byte for byte; 1 GiB streams of other kinds: decoded segment-wise?
(timeout 400 python tools/gpu_big_buffer.py --mib 4100 2>/dev/null | tail -1
 for k in "--kind text" "--level -2" "--level 6" "--kind rand"; do timeout 200 python tools/gpu_big_buffer.py --mib 1024 --no-oracle --no-zlib $k 2>/dev/null | tail -1; done) > $O/${T}_big_buffer.json
ZH_L1_PARSE=parallel timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1 > $O/${T}_single_call_parallel_parse.json
ZH_L1_PARSE=parallel timeout 300 python bench.py --buffers 512 --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --no-parity-sample 2>/dev/null | tail -1 > $O/${T}_share512_parallel_parse.json
timeout 1200 python tools/gpu_fuzz.py 1000 120 2>&1 | tail -4 > $O/${T}_fuzz.log
timeout 900 python tools/gpu_fuzz_chain.py 600 20 2>&1 | tail -2 > $O/${T}_fuzz_chain.log
(timeout 900 python tools/gpu_fuzz.py --mutations 10000 2>&1 | tail -3; timeout 900 python tools/gpu_fuzz.py --seg-mutations 4000 2>&1 | tail -2) > $O/${T}_fuzz_damaged.log 2>&1
timeout 300 python tools/kprof.py --l1-parse 1 --buffers 1024 2>&1 | head -13 > $O/${T}_kprof_l1p.txt
timeout 300 python tools/kprof.py --foreign 6 --buffers 1024 > $O/${T}_kprof_foreign6.txt 2>&1
timeout 300 python tools/kprof.py --buffers 1024 > $O/${T}_kprof_exact.txt 2>&1   # (with the sub-phases of the Huffman replay)
cd /tmp
rm -rf /tmp/kt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $O/${T}_rocprof_bench.log 2>&1
cd $R
python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${T}_kernel_stats.csv 2>$O/${T}_summary.err
tail -2 $O/${T}_pytest_gpu.log; for f in bench c4 share512; do echo "== $f"; cut -c1-600 $O/${T}_$f.json; done; head -14 $O/${T}_kernel_stats.csv
