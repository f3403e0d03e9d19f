#!/bin/bash
# Per-kernel PMC collection for the three hot kernels (run on the GPU box through gpurun).
# Launch shape = the timed one: PMC_PAIRS (default 512) pairs per launch, as in `python bench.py`.
# Counters are collected in separate passes (SQ: 8 slots, TCC: FETCH_SIZE=3 + WRITE_SIZE=2 > 4).
set -u
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc}
EXTRA=${2:-}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --pairs ${PMC_PAIRS:-512} --steps 1 --warmup 1 --cpu-pairs 0 --no-profile --skip-secondary --pairs-in-flight ${PMC_PAIRS:-512} $EXTRA"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $CMD > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -R "$OUT" | head -40
