#!/bin/bash
set -e
R=$PWD; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_dw_i -- python3 $R/bench/experiments/dw_one.py "$@" > $R/gpurun_out/pmc_dw_i.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_dw_b -- python3 $R/bench/experiments/dw_one.py "$@" > $R/gpurun_out/pmc_dw_b.log 2>&1
