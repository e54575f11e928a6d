#!/bin/bash
# GPU-BOX TOOLING (scratch): SQ counters of the two K1a kernels in one process
set -o pipefail
OUT=gpurun_out/prof_k1a
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 tools/bench_ab.py --knob k1a_three --values 0 1 --rounds 2 --evals 2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/a -- $CMD > /dev/null 2> $OUT/a.err
echo a done
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH --output-format csv -d $OUT/b -- $CMD > /dev/null 2> $OUT/b.err
echo b done
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/c -- $CMD > /dev/null 2> $OUT/c.err
echo c done
python3 tools/pmc_counters.py $OUT/a $OUT/b $OUT/c > gpurun_out/r05_pmc_k1a_two_vs_three.json
