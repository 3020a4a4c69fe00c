# more seeds on the round's final sources (on the GPU box): bash tools/prof/r04_fuzz_more.sh
O=gpurun_out; mkdir -p $O
(timeout 1500 python tools/gpu_fuzz.py 7000 250 2>&1 | tail -2
 timeout 900 python tools/gpu_fuzz_chain.py 900 40 2>&1 | tail -1
 timeout 1200 python tools/gpu_fuzz.py --mutations 30000 2>&1 | tail -3
 timeout 900 python tools/gpu_fuzz.py --seg-mutations 8000 2>&1 | tail -2) > $O/r04_j_fuzz_more.log 2>&1
cat $O/r04_j_fuzz_more.log
