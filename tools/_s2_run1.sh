cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "dcn or deform" > gpurun_out/s2_pytest_dcn.txt 2>&1; tail -3 gpurun_out/s2_pytest_dcn.txt
export PRN_BENCH_NO_FP32_RUN=1
for v in 0 1 0 1; do PRN_DCN_CSR1=$v python bench.py --no-exchange-probe --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --dcn-offsets 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('CSR1=$v',d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -o t -- python $GRAFT_REPO_ROOT/bench.py --no-exchange-probe --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --dcn-offsets 0 > /tmp/sg.log 2>&1
python $GRAFT_REPO_ROOT/tools/stream_gaps.py /tmp/sg 3 > $GRAFT_REPO_ROOT/gpurun_out/s2_stream_gaps.txt 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/s2_stream_gaps.txt
