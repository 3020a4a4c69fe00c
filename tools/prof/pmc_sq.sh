#!/bin/bash
# Scheduler (SQ) and L2 (TCC) counters per kernel, for the statements "bound by ..." in DESIGN.md §4: rocprofv3 --pmc
# passes (--kernel-trace only, counters in their own runs as the pool's rules ask) of three commands -- the headline step
# with both parses, one GPU's share at DefaultCompression, zlib level-6 members -- summarised per kernel by
# tools/prof/pmc_sq_summary.py.  Run on the GPU box from the repo root:
#   bash tools/prof/pmc_sq.sh <tag> [buffers] [workload ...]     -> gpurun_out/<tag>_pmc_sq.txt (copy it to profiles/)
# The TCC request counters by size (32 / 64 / 128 B: gfx950 has them) give the bytes that crossed the L2's memory side
# exactly, whatever the access pattern; FETCH_SIZE (which tallies 128-byte requests at 64 on this stack) is what
# tools/prof/pmc_passes.sh keeps measuring for hbm_traffic.json, and the two are compared in the summary.
R=$(pwd); T=${1:-r05}; N=${2:-1024}; shift; shift
O=$R/gpurun_out; mkdir -p $O/pmc_$T
export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-parity-sample"
declare -A CMD
CMD[headline]="$B --buffers $N"
CMD[c4_share]="$B --buffers 512 --level -1 --compress-only --no-parallel-parse"
CMD[c3_zlib6]="$B --buffers $N --foreign 6"
CMD[l1]="$B --buffers $N --compress-only --no-parallel-parse"        # the exact BestSpeed matcher's pass alone
CMD[c3_own]="$B --buffers $N --uncompress-only --no-parallel-parse"  # the inflate pair on this library's streams
W=${@:-headline c4_share c3_zlib6}
cd /tmp
pass() {  # workload, pass name, counters...
  w=$1; p=$2; shift; shift
  rm -rf /tmp/sq_${w}_$p
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/sq_${w}_$p -o $p -- ${CMD[$w]} > $O/pmc_$T/${w}_$p.log 2>&1
  echo "$w $p rc=$?"
}
for w in $W; do
  pass $w p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
  pass $w p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
  pass $w p3 SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VALU
  pass $w p4 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  pass $w p5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum
  pass $w p6 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_EA0_WRREQ_DRAM_sum
  pass $w p7 FETCH_SIZE
done
cd $R
for w in $W; do
  echo "==== $w: ${CMD[$w]#python $R/}"
  python tools/prof/pmc_sq_summary.py $(for p in p1 p2 p3 p4 p5 p6 p7; do echo /tmp/sq_${w}_$p; done)
done > $O/${T}_pmc_sq.txt 2>&1
grep -E "^(====|zh_|   (valu|lane|wait|l2 hit|bytes|salu))" $O/${T}_pmc_sq.txt | head -150
