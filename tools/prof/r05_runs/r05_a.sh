export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_a_pytest_gpu.log 2>&1; tail -2 $O/r05_a_pytest_gpu.log
echo "== own"; bash tools/prof/r05_ab.sh "_base - _tok128" --no-parallel-parse 2>&1 | tee $O/r05_a_ab_own.txt
echo "== zlib6"; bash tools/prof/r05_ab.sh "_base - _tok128" --foreign 6 2>&1 | tee $O/r05_a_ab_zlib6.txt
bash tools/prof/pmc_sq.sh r05_a 1024 > $O/r05_a_pmc_sq.log 2>&1; tail -5 $O/r05_a_pmc_sq.log
