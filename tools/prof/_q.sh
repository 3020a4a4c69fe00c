for nb in 64 512 2048; do
echo "== amax4 nb=$nb"
timeout 300 python bench.py --buffers $nb --level -1 --compress-only --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parallel-parse 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['ratio'], {k:v for k,v in d['kernels_ms'].items() if 'chain' in k})
    elif 'fault' in l or 'rror' in l: print(l[:300])
"
done
