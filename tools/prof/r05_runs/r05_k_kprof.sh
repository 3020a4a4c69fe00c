export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python tools/kprof.py --l1-parse 1 --buffers 1024 2>&1 | head -13 > $O/r05_k_kprof_l1p.txt
timeout 300 python tools/kprof.py --foreign 6 --buffers 1024 > $O/r05_k_kprof_foreign6.txt 2>&1
timeout 300 python tools/kprof.py --buffers 1024 > $O/r05_k_kprof_exact.txt 2>&1
tail -14 $O/r05_k_kprof_exact.txt | head -12; grep -A9 "tokens_kernel (thread" $O/r05_k_kprof_foreign6.txt
