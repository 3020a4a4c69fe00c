# round 6, pass ah: contract mode's matcher leaves the coverage bitmap too, this time without LDS traffic: a thread's word of
# the bitmap is its chunk up to its entry (inside a match that began before) + the inside of its own matches -- registers and
# one store a thread.  Whole step, both parses; main lib = the sources before.
export TMPDIR=/tmp
run() { ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip$1.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); pp=d.get('parallel_parse') or {}; print('lib$1', d['value'], d['ms_per_step'], (d.get('parity_sample') or {}).get('identical'), '| contract', d.get('value_parallel_parse'), pp.get('ms_per_step'), (pp.get('parity_sample') or {}), {k:round(v,3) for k,v in (pp.get('kernels_ms') or {}).items() if v > 4.0})"; }
for rep in 1 2 3; do run ""; run _pcov; done
echo "== one GPU's share"
for rep in 1 2; do run "" --buffers 512; run _pcov --buffers 512; done
