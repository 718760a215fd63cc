#!/bin/bash
set -e
R=$PWD; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_gemm8 -- python3 $R/bench/experiments/gemm8_one.py > $R/gpurun_out/pmc_gemm8.log 2>&1
