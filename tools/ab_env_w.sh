#!/bin/bash
# like ab_env.sh for another workload:  WORKLOAD=c1 tools/ab_env_w.sh - "A=1"
for rep in 1 2; do
  for e in "$@"; do
    [ "$e" = "-" ] && e=""
    echo "== ${WORKLOAD:-c1} [$e] $(env $e python bench.py --workload ${WORKLOAD:-c1} --no-roofline --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
