timeout 900 python -m pytest tests -m gpu -x -q -k "parallel or smoke" 2>&1 | tail -2
ZH_L1_PARSE=parallel bash tools/prof/r04_ab.sh _base --compress-only --no-parallel-parse
timeout 300 python tools/kprof.py --l1-parse 1 --buffers 1024 2>&1 | grep -A10 "== zh_l1p_match" | head -12
