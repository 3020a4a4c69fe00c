#!/bin/bash
# Memory-path counters per kernel -- address translation (UTCL1), the L1's request latency towards the L2, the L2's towards
# the fabric (LEVEL / requests = average latency in cycles, Little's law), the texture-address unit's busy and stall cycles --
# for the question "what does a step of the exact matcher wait for".  rocprofv3 --pmc passes with --kernel-trace only.
#   bash tools/prof/pmc_mem.sh <tag> [buffers] [workload ...]   -> gpurun_out/<tag>_pmc_mem.txt
R=$(pwd); T=${1:-r06}; N=${2:-1024}; shift; shift
O=$R/gpurun_out; mkdir -p $O/pmcm_$T
export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-parity-sample"
declare -A CMD
CMD[l1]="$B --buffers $N --compress-only --no-parallel-parse"
CMD[l1p]="$B --buffers $N --compress-only"
CMD[c3_own]="$B --buffers $N --uncompress-only --no-parallel-parse"
CMD[c4_share]="$B --buffers 512 --level -1 --compress-only --no-parallel-parse"
W=${@:-l1}
cd /tmp
pass() {
  w=$1; p=$2; shift; shift
  rm -rf /tmp/pm_${w}_$p
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pm_${w}_$p -o $p -- ${CMD[$w]} > $O/pmcm_$T/${w}_$p.log 2>&1
  echo "$w $p rc=$?"
}
for w in $W; do
  pass $w m1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum GRBM_GUI_ACTIVE
  pass $w m2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
  pass $w m3 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
  pass $w m4 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
  pass $w m5 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum
  pass $w m6 TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pass $w m7 TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
done
cd $R
for w in $W; do
  echo "==== $w: ${CMD[$w]#python $R/}"
  python tools/prof/pmc_sq_summary.py $(for p in m1 m2 m3 m4 m5 m6 m7; do echo /tmp/pm_${w}_$p; done)
done > $O/${T}_pmc_mem.txt 2>&1
grep -E "^(====|zh_|      )" $O/${T}_pmc_mem.txt | grep -v "^zh_\(huff\|layout\|trailer\|unwrap\|verify\|checksum_comb\|l1_cost\|l1_set\)" | head -120
