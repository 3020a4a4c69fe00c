# more seeds on the round's final sources (on the GPU box): bash tools/prof/r04_fuzz_more.sh
O=gpurun_out; mkdir -p $O
(timeout 110 python tools/gpu_fuzz.py 7000 150 2>&1 | tail -2
 timeout 60 python tools/gpu_fuzz_chain.py 900 20 2>&1 | tail -1
 timeout 70 python tools/gpu_fuzz.py --seg-mutations 8000 2>&1 | tail -2
 timeout 60 python tools/gpu_big_buffer.py --mib 2048 --no-oracle --no-zlib --level 9 2>&1 | tail -1 | cut -c1-300) > $O/r04_o_fuzz_more.log 2>&1
cat $O/r04_o_fuzz_more.log
