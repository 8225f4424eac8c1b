#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
L=tools/native/gemm_lab.bin
run() { echo "### $*"; if [ -n "$F" ]; then env "$@" timeout 60 $L "$F"; else env "$@" timeout 60 $L; fi; }
{
F="tail" run PRN_PK_V=3
F="" run PRN_PK_V=3
} > gpurun_out/lab_k.txt 2>&1
