export TMPDIR=/tmp
# (round 5: ZH_L1_TABLE=lds exists only in the test build: python -m zippy_amd.build --variant xcheck -DZH_XCHECK)
export ZIPPY_HIP_LIB=$(pwd)/zippy_amd/libzippy_hip_xcheck.so
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
ZH_L1_TABLE=lds python bench.py --steps 3 --warmup 1 --no-cpu-baseline --compress-only 2>/dev/null | tail -1 > $O/r02_ldsmode_bench.json
python -c "
import json;d=json.load(open('$O/r02_ldsmode_bench.json'));print('lds mode', d['compress_GiBps'], d['kernels_ms'])"
cd /tmp; rm -rf /tmp/pf2 /tmp/pw2
ZH_L1_TABLE=lds timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf2 -o pf -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --compress-only > /dev/null 2>&1
ZH_L1_TABLE=lds timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw2 -o pw -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --compress-only > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $(find /tmp/pf2 -name "*.db" | head -1) $(find /tmp/pw2 -name "*.db" | head -1) --buffers 4096 --size 1048576 > $O/r02_ldsmode_hbm_traffic.json
python -c "
import json;t=json.load(open('$O/r02_ldsmode_hbm_traffic.json'));k=t['kernels'];print({n:(round(v['hbm_bytes_per_launch']/1e9,2),round(v['hbm_bytes_per_launch_uncorrected']/1e9,2)) for n,v in k.items() if v['hbm_bytes_per_launch']>1e8})"
