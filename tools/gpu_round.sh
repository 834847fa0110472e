#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, HBM traffic counters.
# usage (from the repo root on the box): bash tools/gpu_round.sh <tag> [steps]
TAG=${1:-r01}
STEPS=${2:-40}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/${TAG}_pytest_gpu.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1
echo "smoke exit $?"; tail -2 $OUT/${TAG}_smoke.log
CREAM_BENCH_EXTRA=$OUT/${TAG}_bench_extra.json timeout 600 python bench.py --steps $STEPS --warmup 10 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?"; cat $OUT/${TAG}_bench.json
cd /tmp
CREAM_BENCH_EXTRA=/dev/null timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o step -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
echo "rocprof exit $?"; cat $OUT/${TAG}_prof_bench.json
# QUICK=1: tests, bench line, kernel stats, the rpe_index HBM counters and config 2 only (the other records are unchanged)
# counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
[ -z "$QUICK" ] && for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  # full-size steps only (no batch-4 host leg, no extra legs): the per-kernel means are those of the benchmarked shapes
  CREAM_BENCH_EXTRA=/dev/null   timeout 600 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'attn_|table_images|gemm_|ln_|adamw|grad_finalize|soft_ce|tail_|stem_' -d $OUT/${TAG}_pmc_$N -o pmc --output-format csv -- python $REPO/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-host-leg --no-kernel-timing > /dev/null 2> $OUT/${TAG}_pmc_$N.err
  echo "pmc $N exit $?"
  # config 4: the fused iRPE attention kernels and the rpe_index kernels under the same counters (bench.py reads their mfma_util / traffic)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'irpe_' -d $OUT/${TAG}_pmc_irpe_$N -o pmc --output-format csv -- python $REPO/tools/bench_irpe_attention.py > /dev/null 2> $OUT/${TAG}_pmc_irpe_$N.err
  timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'rpe_gather|rpe_scatter' -d $OUT/${TAG}_pmc_rpe_$N -o pmc --output-format csv -- python $REPO/tools/bench_rpe_index.py > /dev/null 2> $OUT/${TAG}_pmc_rpe_$N.err
done
# BASELINE config 4: rocprofv3 kernel trace + HBM counters of the rpe_index kernels, stall counters of both kernel families
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_rpe_prof -o rpe -- python $REPO/tools/bench_rpe_index.py > /dev/null 2> $OUT/${TAG}_rpe_prof.err
i=0
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'rpe_gather|rpe_scatter' -d $OUT/${TAG}_rpepmc_$i -o pmc --output-format csv -- python $REPO/tools/bench_rpe_index.py > /dev/null 2> $OUT/${TAG}_rpepmc_$i.err
  echo "rpe pmc $i exit $?"
  [ -n "$QUICK" ] && [ $i -ge 2 ] && break
  [ $i -ge 3 ] && timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'irpe_attn' -d $OUT/${TAG}_irpepmc_$i -o pmc --output-format csv -- python $REPO/tools/bench_irpe_attention.py > /dev/null 2> $OUT/${TAG}_irpepmc_$i.err
done
cd $REPO
# the other measured configurations: config 2 (supernet-T, fixed subnet), config 4 (rpe_index
# micro-benchmark + one RPEAttention layer), sub-network evaluation, host-side profile of the step
CREAM_BENCH_EXTRA=$OUT/${TAG}_bench_config2_extra.json timeout 300 python bench.py --supernet T --subnet T --steps 40 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_config2.json 2> $OUT/${TAG}_bench_config2.err
echo "config2 exit $?"; cat $OUT/${TAG}_bench_config2.json | cut -c1-400
timeout 300 python tools/bench_rpe_index.py > $OUT/${TAG}_rpe_index_microbench.jsonl 2> $OUT/${TAG}_rpe_index_microbench.err
echo "rpe microbench exit $?"; cut -c1-300 $OUT/${TAG}_rpe_index_microbench.jsonl
if [ -z "$QUICK" ]; then
timeout 300 python tools/bench_irpe_attention.py > $OUT/${TAG}_irpe_attention.jsonl 2> $OUT/${TAG}_irpe_attention.err
echo "irpe attention exit $?"; cut -c1-300 $OUT/${TAG}_irpe_attention.jsonl
timeout 300 python tools/bench_subnet_eval.py > $OUT/${TAG}_subnet_eval.json 2> $OUT/${TAG}_subnet_eval.err
echo "subnet eval exit $?"; cut -c1-300 $OUT/${TAG}_subnet_eval.json
timeout 300 python tools/host_profile.py > $OUT/${TAG}_host_profile.txt 2>&1
echo "host profile exit $?"; head -12 $OUT/${TAG}_host_profile.txt
fi
find $OUT -name '*.csv' -path "*${TAG}*" | head -20
# idle time between the kernels of each queue (needs the raw trace: before it is dropped)
TRACE=$(find $OUT -name '*kernel_trace.csv' -path "*${TAG}_prof*" | head -1)
[ -n "$TRACE" ] && python $REPO/tools/summarize_gaps.py $TRACE > $OUT/${TAG}_step_gaps.txt 2>&1 && head -12 $OUT/${TAG}_step_gaps.txt
# keep the merge small: drop raw traces, keep stats + counter csv
find $OUT -name '*kernel_trace.csv' -path "*${TAG}_prof*" -delete
find $OUT -name '*.db' -delete
du -sh $OUT
