cd $GRAFT_REPO_ROOT
for t in 0 2097152 0 2097152; do
  PRN_EXCHANGE_TAIL_BYTES=$t PRN_BENCH_NO_FP32_RUN=1 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --dcn-offsets 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tail', $t, 'ms', round(d['ms_per_step'],2), d['exchange_probe'])"
done
