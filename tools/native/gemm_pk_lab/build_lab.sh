#!/bin/bash
# builds the library and the lab binary; prints only errors
cd "$(dirname "$0")/../.."
python planerecnet_amd/build.py > /tmp/build_lib.log 2>&1 || { grep -i error -A5 /tmp/build_lib.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/native/gemm_lab.cpp -o tools/native/gemm_lab.bin -Lplanerecnet_amd -lprn_hip -Wl,-rpath,'$ORIGIN/../../planerecnet_amd' > /tmp/build_lab.log 2>&1 || { cat /tmp/build_lab.log; exit 1; }
echo built
