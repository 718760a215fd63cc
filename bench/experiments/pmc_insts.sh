#!/bin/bash
# usage: pmc_insts.sh <tag> <one_layer args...>   -- instruction-mix counter pass, CSV under gpurun_out/pmc_<tag>_i
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_${tag}_i -- python3 bench/one_layer.py "$@" > gpurun_out/pmc_${tag}_i.log 2>&1
