#!/bin/bash
set -o pipefail
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_distributed.py -x -v --timeout 180 --timeout-method thread > $O/peer_tests.log 2>&1
rc=$?
tail -40 $O/peer_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "relu_tag or premasked or masked_data" --timeout 120 --timeout-method thread 2>&1 | tail -5
