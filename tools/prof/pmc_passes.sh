#!/bin/bash
# Memory-side traffic of the headline step AND of BASELINE configs 2-5, each from its own rocprofv3 PMC passes
# (FETCH_SIZE, WRITE_SIZE and the read requests by size in separate runs; --kernel-trace only, as the pool's rules ask), merged into
# gpurun_out/hbm_traffic.json by tools/pmc_traffic.py.  Run on the GPU box from the repo root:
#   bash tools/prof/pmc_passes.sh [workload ...]        (default: all)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-parity-sample"
declare -A CMD KNOWN ENVV
CMD[headline]="$B";                                                                     KNOWN[headline]=$((4096*1048576))
CMD[c2]="$B --buffers 1024 --size 65536 --compress-only --no-parallel-parse";          KNOWN[c2]=$((1024*65536))
CMD[c2_parallel_parse]="$B --buffers 1024 --size 65536 --compress-only --no-parallel-parse"; KNOWN[c2_parallel_parse]=$((1024*65536)); ENVV[c2_parallel_parse]="ZH_L1_PARSE=parallel"
CMD[c3_zlib6]="$B --foreign 6";                                                         KNOWN[c3_zlib6]=$((4096*1048576))
CMD[c4_share]="$B --buffers 512 --level -1 --compress-only --no-parallel-parse";       KNOWN[c4_share]=$((512*1048576))
CMD[c5]="python $R/tools/bench_c5.py --steps 1";                                        KNOWN[c5]=$((128*1048576))
W=${@:-headline c2 c2_parallel_parse c3_zlib6 c4_share c5}
rm -f $O/hbm_traffic.json
cd /tmp
for w in $W; do
  rm -rf /tmp/pf_$w /tmp/pw_$w /tmp/ps_$w
  env ${ENVV[$w]} timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$w -o pf -- ${CMD[$w]} > /dev/null 2>$O/pmc_$w.err
  env ${ENVV[$w]} timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$w -o pw -- ${CMD[$w]} > /dev/null 2>>$O/pmc_$w.err
  env ${ENVV[$w]} timeout 500 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/ps_$w -o ps -- ${CMD[$w]} > /dev/null 2>>$O/pmc_$w.err
  (cd $R && python tools/pmc_traffic.py $(find /tmp/pf_$w -name "*.db" | head -1) $(find /tmp/pw_$w -name "*.db" | head -1) \
      --sized-db "$(find /tmp/ps_$w -name "*.db" | head -1)" \
      --workload $w --known-bytes ${KNOWN[$w]} --command "${ENVV[$w]} ${CMD[$w]#python $R/}" --merge-into $O/hbm_traffic.json) 2>>$O/pmc_$w.err
done
