cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -k "k_split or batch_norm or bottleneck_chain or winograd" > gpurun_out/s2_pytest_lazy.txt 2>&1; tail -5 gpurun_out/s2_pytest_lazy.txt
export PRN_BENCH_NO_FP32_RUN=1
for v in 0 1 0 1; do PRN_LAZY_SPLIT_SUM=$v python bench.py --no-exchange-probe --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --dcn-offsets 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('LAZY=$v',d['ms_per_step'])"; done
python -m pytest tests/test_model_gpu.py tests/test_r101_train_gpu.py -x -q > gpurun_out/s2_pytest_model.txt 2>&1; tail -5 gpurun_out/s2_pytest_model.txt
