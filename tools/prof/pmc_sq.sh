#!/bin/bash
# Instruction-mix / stall counters per kernel (several rocprofv3 --pmc passes of a small bench).
# Usage (on the GPU box, from the repo root): bash tools/prof/pmc_sq.sh [buffers]
R=$(pwd); N=${1:-1024}
export TMPDIR=/tmp; cd /tmp
mkdir -p $R/gpurun_out/pmc
pass() {
  name=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc/$name -o $name -- \
    python $R/bench.py --buffers $N --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc/$name.log 2>&1
  echo "$name rc=$?"
}
pass p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass p2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
pass p4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
pass p5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass p6 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
cd $R
python tools/prof/pmc_sq_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt 2>&1
tail -60 gpurun_out/pmc/summary.txt
