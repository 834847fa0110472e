cd /tmp && export TMPDIR=/tmp
python /root/repo/tools/bench_image_transform.py train
python /root/repo/tools/bench_image_transform.py eval
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/img_prof -o img -- python /root/repo/tools/bench_image_transform.py train > /dev/null 2>&1
grep -h "image_resample" /root/repo/gpurun_out/img_prof/*kernel_stats.csv | cut -c1-200
timeout 600 python -m pytest /root/repo/tests/test_image_transform_gpu.py -q 2>&1 | tail -2
