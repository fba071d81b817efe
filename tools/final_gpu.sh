# Round-end GPU pass (run on the GPU box): full -m gpu suite, smoke, profile round, bench lines of the other configurations
set -e
TAG=${1:-r02}
mkdir -p gpurun_out/final_$TAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_$TAG/gpu_tests.log 2>&1 || { tail -30 gpurun_out/final_$TAG/gpu_tests.log; exit 1; }
tail -2 gpurun_out/final_$TAG/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_$TAG/smoke.log 2>&1; tail -1 gpurun_out/final_$TAG/smoke.log
bash tools/profile_round.sh $TAG > gpurun_out/final_$TAG/profile.log 2>&1; tail -c 150 gpurun_out/final_$TAG/profile.log; echo
