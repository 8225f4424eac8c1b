cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r05_c_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r05_c_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/final_profiles.sh r05_c > gpurun_out/r05_c_final.log 2>&1; tail -5 gpurun_out/r05_c_final.log
