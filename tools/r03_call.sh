cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03b_gpu.log 2>&1
echo "gpu tests exit $?"; grep -E "^\[fp32|passed|failed|Error|error|assert" gpurun_out/r03b_gpu.log | tail -40
