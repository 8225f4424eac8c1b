#!/bin/bash
# builds the lab binary against the product library (LAB_NO_PK: "old" column only).  To time the lab kernels, copy
# prn_gemm_pk.hip into planerecnet_amd/csrc/ (it defines prn_gemm_kn), rebuild the library and drop -DLAB_NO_PK.
cd "$(dirname "$0")/../../.."
python planerecnet_amd/build.py > /tmp/build_lib.log 2>&1 || { grep -i error -A5 /tmp/build_lib.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value ${LAB_FLAGS--DLAB_NO_PK} tools/native/gemm_pk_lab/gemm_lab.cpp -o tools/native/gemm_pk_lab/gemm_lab.bin -Lplanerecnet_amd -lprn_hip -Wl,-rpath,'$ORIGIN/../../../planerecnet_amd' > /tmp/build_lab.log 2>&1 || { cat /tmp/build_lab.log; exit 1; }
echo built
