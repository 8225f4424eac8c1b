#!/bin/bash
# Host-side load of an N-rank job measured on ONE GPU.  N copies of tools/host_vs_gpu.py run at the same time (HOST_ONLY=1): each enqueues whole
# training steps of PlaneRecNet_101 (per-rank batch 8) while ITS stream is parked behind a spin kernel, so what is timed is the host -- wall time a
# rank needs to enqueue a step and the CPU time of its process -- under the contention of N ranks on the node's cores.  (Letting N ranks actually
# share the device measures the driver's time slicing instead: 7 s per step with 8 ranks.)   BATCH=2 tools/host_probe_8ranks.sh [N=8]
# (BATCH=2 per rank keeps eight resident training steps far below the HBM size; the launches a rank enqueues per step do not depend on the batch)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-8}
cd $R
O=gpurun_out/host_probe_${N}ranks.txt
echo "# $N concurrent ranks' host work against one parked MI355X; host: $(nproc) logical CPUs; load before: $(cut -d' ' -f1-3 /proc/loadavg)" > $O
for n in 1 $N; do
  echo "## $n rank(s) at once" >> $O
  for i in $(seq 1 $n); do HOST_ONLY=1 timeout 400 python tools/host_vs_gpu.py > /tmp/hostprobe_$i.txt 2>&1 & done
  wait
  for i in $(seq 1 $n); do echo "rank $i: $(grep -E 'host enqueue' /tmp/hostprobe_$i.txt) | $(grep -E 'process CPU' /tmp/hostprobe_$i.txt)" >> $O; done
done
cat $O
