cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_h_env_matrix2.txt; : > $O
run() { echo "=== $*" >> $O; env "$@" python tools/wait_probe.py 2>&1 | grep -E "free-running|host phases|tid|event.synchronize at _context" | head -8 >> $O; }
run FIXED_TARGETS=1
run FIXED_TARGETS=1 NO_ASYNC=1
run FIXED_TARGETS=1 AMD_DIRECT_DISPATCH=0
run FIXED_TARGETS=1 PRN_BLOCKS=0
cat $O
