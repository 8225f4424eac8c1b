#!/bin/bash
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out
L=tools/native/gemm_split_lab/split_lab.bin
{
for v in ${VERS-1}; do for d in ${DBGS-0}; do for n in ${NPRODS-6}; do echo "### SG_V=$v SG_DBG=$d"; SG_V=$v SG_DBG=$d LAB_NPROD=$n timeout 120 $L $F; done; done; done
} > gpurun_out/split_lab.txt 2>&1
