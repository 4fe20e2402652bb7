set -x
O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests/test_encoder_gpu.py tests/test_prefilter_gpu.py tests/test_pretrained_path.py -m gpu -x -q -s -k "small_batches or prefilter or pretrained or toy or full_forward" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?" >> $O/bench20.err
tail -5 $O/tests.log; tail -3 $O/bench20.err
