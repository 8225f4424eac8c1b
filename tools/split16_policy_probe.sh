#!/bin/bash
# rocprofv3 kernel durations of split16 shapes under each store policy
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in 0 1 3; do for pol in 0 1 2 3 4; do
  rm -rf /tmp/pp; S16_SHAPE=$shape PRN_SPLIT_STORE_POLICY=$pol rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python $R/tools/split16_policy_probe.py > /dev/null 2>&1
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  echo "shape $shape policy $pol: $(grep -i "split16_gemm_kernel\|relu\|clamp" $f | awk -F, '{printf "%s calls %s avg %.1f us | ", substr($1,1,40), $2, $4/1000}')"
done; done
