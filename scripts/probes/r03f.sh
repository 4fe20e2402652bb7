set -x
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_prefilter_gpu.py tests/test_search_gpu.py tests/test_configs_gpu.py tests/test_subnormal_bound.py tests/test_cabi.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
S="100000,64,10 100000,32,10 100000,1,10 1000000,64,10 100000,64,100 1000000,1,50"
timeout -k 10 300 python scripts/probes/search_bench.py $S > $O/search_plain.log 2>&1
UNIT=1 timeout -k 10 300 python scripts/probes/search_bench.py $S > $O/search_unit.log 2>&1
cat $O/search_plain.log $O/search_unit.log | grep rows | cut -c1-330
timeout -k 10 200 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "small_batches" > $O/enc_tests.log 2>&1; tail -3 $O/enc_tests.log
