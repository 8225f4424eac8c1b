#!/bin/bash
# Where do the waves of split16_gemm_kernel wait?  Four PMC passes over one shape (S16_SHAPE=0: 1024 <- 256 @30x40 + residual, B = 8).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmcs_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcs_$i -o t -- python $R/tools/split16_one_shape.py > /tmp/pmcs_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
c = collections.defaultdict(float)
n = 0
for i in (1, 2, 3, 4):
    for f in glob.glob('/tmp/pmcs_%d/*counter_collection.csv' % i):
        for r in csv.DictReader(open(f)):
            if 'split16_gemm_kernel' in r['Kernel_Name']:
                c[r['Counter_Name']] += float(r['Counter_Value'])
g = lambda k: c.get(k, 0.0)
print("split16_gemm_kernel")
print("   LDS bank conflict cycles / LDS active cycles: %.3f   (LDS instructions %.3g)" % (g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1), g('SQ_INSTS_LDS')))
wc = max(g('SQ_WAVE_CYCLES'), 1)
print("   of the wave cycles: waiting on any instruction %.2f, on LDS %.2f, waiting for anything %.2f" % (g('SQ_WAIT_INST_ANY') / wc, g('SQ_WAIT_INST_LDS') / wc, g('SQ_WAIT_ANY') / wc))
bc = max(g('SQ_BUSY_CYCLES'), 1)
print("   issue activity / busy cycles: VALU(+MFMA) %.2f  VMEM %.2f  LDS %.2f  scalar %.2f  misc %.2f  any %.2f ; MFMA busy %.2f" % (
    g('SQ_ACTIVE_INST_VALU') / bc, g('SQ_ACTIVE_INST_VMEM') / bc, g('SQ_ACTIVE_INST_LDS') / bc, g('SQ_ACTIVE_INST_SCA') / bc, g('SQ_ACTIVE_INST_MISC') / bc,
    g('SQ_ACTIVE_INST_ANY') / bc, g('SQ_VALU_MFMA_BUSY_CYCLES') / bc))
print("   raw:", {k: "%.3g" % v for k, v in sorted(c.items())})
PY
