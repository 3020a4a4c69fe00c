run() { env $1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs 2>gpurun_out/err14.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('value_parallel_parse'), 'c', d['compress_GiBps'], {k:round(v,2) for k,v in d['kernels_ms'].items() if v > 0.25}, d.get('kernel_launches'), d['parity_sample']['identical'])"; tail -1 gpurun_out/err14.txt; }
run ZH_COMPRESS_CHUNKS=1
run ZH_COMPRESS_CHUNKS=4
run "ZH_COMPRESS_CHUNKS=4 ZH_L1_SLOTS=4096"
run ZH_COMPRESS_CHUNKS=8
run "ZH_COMPRESS_CHUNKS=8 ZH_L1_SLOTS=4096"
run ZH_COMPRESS_CHUNKS=2
