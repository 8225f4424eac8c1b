#!/bin/bash
# which part of the split kernel makes the firmware lower the shader clock?  Sustained launches of one kernel, sclk / power sampled.
cd "$(dirname "$0")/../../.."
L=tools/native/gemm_split_lab/split_lab.bin
probe() {
  ( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | grep card0; sleep 0.2; done ) > /tmp/smi_probe.csv &
  SMI=$!
  env "$@" LAB_REPS=${REPS-500} timeout 120 $L "${F-4096}" | grep -E "4096|fpn|s3 256" | cut -c1-75
  kill $SMI; wait $SMI 2>/dev/null
  python3 - "$*" <<'PY'
import sys
rows = [l.strip().split(",") for l in open("/tmp/smi_probe.csv") if l.strip()]
busy = [(int(r[5].strip("()Mhz")), float(r[-1])) for r in rows if float(r[-1]) > 450]
busy = busy[len(busy) // 3:]
if busy:
    print("   %-40s sclk mean %.0f MHz (min %d), power %.0f W, %d samples" % (sys.argv[1], sum(b[0] for b in busy) / len(busy), min(b[0] for b in busy), sum(b[1] for b in busy) / len(busy), len(busy)))
PY
}
probe PRN_SPLIT_GEMM=0 LAB_ONLY=old
probe LAB_ONLY=split SG_V=1 LAB_NPROD=6
probe PRN_SPLIT_GEMM=0 LAB_ONLY=old
probe LAB_ONLY=split SG_V=1 LAB_NPROD=6
