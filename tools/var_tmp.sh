for v in 0 1 2 16 19 4 8; do echo "== var $v"; ZH_L1P_VAR=$v python tools/kprof.py --l1-parse 1 --buffers 512 2>&1 | grep -A7 "zh_l1p_match_kernel (thread" | grep "P1\|P2\|P3\|P4\|total"; done
