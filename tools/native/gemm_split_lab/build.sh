#!/bin/bash
# builds the lab binary (kernels in split_gemm.h) against the product library for the "old" column
cd "$(dirname "$0")/../../.."
D=tools/native/gemm_split_lab
mkdir -p /tmp/sg_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -save-temps=obj -o /tmp/sg_tmp/split_lab.bin $D/split_lab.cpp -Lplanerecnet_amd -lprn_hip -Wl,-rpath,'$ORIGIN/../../../planerecnet_amd' > /tmp/sg_build.log 2>&1 || { cat /tmp/sg_build.log; exit 1; }
cp /tmp/sg_tmp/split_lab.bin $D/split_lab.bin
grep -E "^_Z.*sg_.*:$|; NumVgprs|; NumAgprs|; ScratchSize|; Occupancy|; LDSByteSize" /tmp/sg_tmp/split_lab-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - - | sed 's/\t/ /g' | grep -v "ref64\|fill_k"
echo built
