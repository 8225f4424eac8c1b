cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r05_e_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r05_e_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/final_profiles.sh r05_e > gpurun_out/r05_e_final.log 2>&1; tail -3 gpurun_out/r05_e_final.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sg; PRN_BENCH_NO_FP32_RUN=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -o t -- python $GRAFT_REPO_ROOT/bench.py --no-exchange-probe --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --dcn-offsets 0 > /tmp/sg.log 2>&1
python $GRAFT_REPO_ROOT/tools/stream_gaps.py /tmp/sg 3 > $GRAFT_REPO_ROOT/gpurun_out/final/r05_e_stream_gaps.txt 2>&1; head -6 $GRAFT_REPO_ROOT/gpurun_out/final/r05_e_stream_gaps.txt
