#!/bin/bash
# the device input transform on the GPU box: tests, benchmark lines, rocprofv3 kernel stats and the HBM counters (separate passes)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python -m pytest $REPO/tests/test_image_transform_gpu.py -q 2>&1 | tail -2
python $REPO/tools/bench_image_transform.py train
python $REPO/tools/bench_image_transform.py eval
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/img_prof -o img -- python $REPO/tools/bench_image_transform.py train > /dev/null 2>&1
grep -h "image_" $OUT/img_prof/*kernel_stats.csv | cut -c1-160
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'image_' -d $OUT/img_pmc_$C -o pmc --output-format csv -- python $REPO/tools/bench_image_transform.py train > /dev/null 2> $OUT/img_pmc_$C.err
  echo "pmc $C exit $?"
done
