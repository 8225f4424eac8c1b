#!/bin/bash
# usage: tools/pmc_run.sh "<shape substring>" <fwd|dgrad|wgrad>   -- prints per-kernel PMC averages (two counter passes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  rm -rf /tmp/pmc1
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/conv_one.py "$1" $2 3 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc1/*counter_collection.csv')
if not f:
    print("no counter file", glob.glob('/tmp/pmc1/*')); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'conv_' not in k: continue
    acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
