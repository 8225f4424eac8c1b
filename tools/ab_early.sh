#!/bin/bash
# within-call A/B: results received by the receiver thread (default) or in line by the trainer
for i in 1 2; do
  for e in 1 0; do
    echo "== PRN_PREFETCH_EARLY=$e"
    PRN_PREFETCH_EARLY=$e PRN_BENCH_GAP=1 PRN_BENCH_PHASES=1 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>&1 | grep -E "ms_per_step|get_wait|GPU time from" | cut -c1-400
  done
done
