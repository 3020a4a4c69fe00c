bash tools/prof/r04_ab.sh _base --compress-only --no-parallel-parse
bash tools/prof/r04_ab.sh _base --compress-only --no-parallel-parse --buffers 1024 --size 65536
for v in "" _base; do ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$v.so timeout 300 python tools/bench_single_call.py 2>/dev/null | tail -1; done
