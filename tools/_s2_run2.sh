cd /tmp && export TMPDIR=/tmp
export PRN_BENCH_NO_FP32_RUN=1
rm -rf /tmp/sg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -o t -- python $GRAFT_REPO_ROOT/bench.py --no-exchange-probe --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --dcn-offsets 0 > /tmp/sg.log 2>&1
python $GRAFT_REPO_ROOT/tools/stream_gaps.py /tmp/sg 3 "FillFunctor|copyBuffer|fillBuffer" > $GRAFT_REPO_ROOT/gpurun_out/s2_stream_gaps2.txt 2>&1
f=$(find /tmp/sg -name "*kernel_trace.csv" | head -1); gzip -c $f > $GRAFT_REPO_ROOT/gpurun_out/s2_kernel_trace.csv.gz; ls -la $GRAFT_REPO_ROOT/gpurun_out/
