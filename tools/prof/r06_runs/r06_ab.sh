# round 6, pass ab: the emission reads the matcher's coverage bitmap (a.f_cover: 4 KiB a fragment, written in the matcher's
# statistics phase) instead of filing the match list into chunk bitmaps a second time.  Arms: the sources before (_base), the new
# ones, the new ones with ZH_EMIT_COVER=0 (the matcher writes the bitmap, the emission ignores it).  Parity sample on.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parallel-parse --compress-only "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$1 cover=${ZH_EMIT_COVER:-1}', d['value'], (d.get('parity_sample') or {}).get('identical'), {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.2})"; }
for rep in 1 2 3; do run _base; run ""; ZH_EMIT_COVER=0 run ""; done
echo "== one GPU's share (512 x 1 MiB)"
for rep in 1 2; do run _base --buffers 512; run "" --buffers 512; done
echo "== config 2 (1024 x 64 KiB)"
for rep in 1 2; do run _base --buffers 1024 --size 65536; run "" --buffers 1024 --size 65536; done
echo "== level -2"
run _base --level -2 --buffers 512; run "" --level -2 --buffers 512
