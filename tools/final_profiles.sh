#!/bin/bash
# The artefact set of a round in one GPU call:  tools/final_profiles.sh r02_j   -> gpurun_out/final/<tag>_*
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py 2>$O/${TAG}_bench_c3.stderr | grep "^{" | tail -1 > $O/${TAG}_bench_c3.json
for w in c1 c2 c5; do python bench.py --workload $w 2>/dev/null | grep "^{" | tail -1 > $O/${TAG}_bench_$w.json; done
bash tools/rocprof_stats.sh $TAG > $O/${TAG}_kernel_stats_summary.txt 2>&1
cp gpurun_out/prof/${TAG}_kernel_stats_serial.csv $O/${TAG}_kernel_stats_serial_sync_wgrad.csv
cp gpurun_out/prof/${TAG}_kernel_stats_default.csv $O/${TAG}_kernel_stats_default_overlapped.csv
timeout 1200 bash tools/pmc_mfma.sh > $O/${TAG}_pmc_mfma.txt 2>&1 && cp gpurun_out/pmc_mfma.json $O/${TAG}_pmc_mfma.json
timeout 2000 bash tools/pmc_traffic.sh > $O/${TAG}_pmc_traffic.txt 2>&1 && cp gpurun_out/pmc_traffic/pmc_traffic.json $O/${TAG}_pmc_traffic.json
python tools/host_vs_gpu.py 2>&1 | grep -E "free-running|host enqueue|GPU drain" > $O/${TAG}_host_vs_gpu.txt
python tools/aten_trace.py 2>/dev/null > $O/${TAG}_aten_ops_by_source_line.txt
PRN_BENCH_GAP=1 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>&1 | grep -E "GPU time|adjacent" > $O/${TAG}_step_boundary.txt
ls -la $O
