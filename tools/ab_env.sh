#!/bin/bash
# within-call A/B of environment settings on the default bench:  tools/ab_env.sh "A=1" "B=2 C=3" ...   ("-" = defaults)
for rep in 1 2; do
  for e in "$@"; do
    [ "$e" = "-" ] && e=""
    echo "== [$e] $(env $e python bench.py --steps ${STEPS:-40} --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
