cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s -k "config4 or whole_step or soft_loss_on_the_device or two_ranks or bf16_autocast" > gpurun_out/r03a_newtests.log 2>&1
echo "new tests exit $?"; grep -E "^\[|passed|failed|Error|error" gpurun_out/r03a_newtests.log | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
echo "bench exit $?"; cat gpurun_out/r03a_bench.json; tail -5 gpurun_out/r03a_bench.err
