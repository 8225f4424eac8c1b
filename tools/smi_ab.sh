#!/bin/bash
# A/B of bench.py configurations on ONE box with the shader clock and socket power sampled four times a second:
#   tools/smi_ab.sh "VAR=a VAR2=b" "VAR=c" ...   (each argument: environment of one run)
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i + 1))
  ( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | grep card0; sleep 0.25; done ) > gpurun_out/smi_$i.csv &
  SMI=$!
  env $cfg python bench.py --steps ${STEPS-60} --warmup 8 --no-cpu-baseline --no-roofline --no-exchange-probe ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
  kill $SMI; wait $SMI 2>/dev/null
  python - <<PY
rows = [l.strip().split(",") for l in open("gpurun_out/smi_$i.csv") if l.strip()]
busy = [(int(r[5].strip("()Mhz")), float(r[-1])) for r in rows if float(r[-1]) > 600]
if busy:
    print("   under load: %d samples, sclk mean %.0f MHz (min %d, max %d), power mean %.0f W" % (len(busy), sum(b[0] for b in busy) / len(busy), min(b[0] for b in busy), max(b[0] for b in busy), sum(b[1] for b in busy) / len(busy)))
PY
done
