#!/bin/bash
# usage: pmc_run.sh <tag> <one_layer args...>   -- two counter passes, CSV under gpurun_out/pmc_<tag>_{a,b}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 bench/one_layer.py "$@" > gpurun_out/pmc_${tag}_a.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 bench/one_layer.py "$@" > gpurun_out/pmc_${tag}_b.log 2>&1
