# round 6, pass ae: the exact matcher's literal counts into 1 (before) / 2 / 4 histograms, a lane to the copy its number picks
# (16-bit counts where the parse kept its slot-written bits): adds to one LDS word are taken one after the other.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run _lit1; run _lit2; run ""; done
echo "== one GPU's share (512 x 1 MiB)"
for rep in 1 2; do run _lit1 --buffers 512; run "" --buffers 512; done
echo "== config 2 (1024 x 64 KiB)"
for rep in 1 2; do run _lit1 --buffers 1024 --size 65536; run "" --buffers 1024 --size 65536; done
