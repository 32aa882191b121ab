#!/bin/bash
# PMC counters for the GEMM micro-benchmark (counters only: no tracing flags besides what rocprofv3 needs).
set -x
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py --quick --iters 3 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/p2 -o p2 -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py --quick --iters 3 > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/p3 -o p3 -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py --quick --iters 3 > $OUT/p3.log 2>&1
ls -R $OUT | head -30
