set -e
mkdir -p gpurun_out/r2final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2final/gpu_tests.log 2>&1 || { tail -30 gpurun_out/r2final/gpu_tests.log; exit 1; }
tail -2 gpurun_out/r2final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2final/smoke.log 2>&1; tail -2 gpurun_out/r2final/smoke.log
bash tools/profile_round.sh r02 > gpurun_out/r2final/profile.log 2>&1; tail -c 200 gpurun_out/r2final/profile.log
