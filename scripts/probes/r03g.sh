set -x
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "skinny or small_batches or full_depth" -s > $O/enc_tests.log 2>&1; echo "rc=$?" >> $O/enc_tests.log; tail -5 $O/enc_tests.log
timeout -k 10 300 python scripts/probes/skinny_bench.py > $O/skinny.log 2>&1; cat $O/skinny.log
timeout -k 10 300 python scripts/probes/small_encode_profile.py > $O/small_encode.log 2>&1; grep "ms per" $O/small_encode.log
timeout -k 10 300 python -m pytest tests/test_prefilter_gpu.py -m gpu -x -q -k "unit_query or broken" > $O/pf_tests.log 2>&1; tail -3 $O/pf_tests.log
S="100000,64,10 100000,32,10 100000,1,10 1000000,64,10"
timeout -k 10 300 python scripts/probes/search_bench.py $S > $O/search_plain.log 2>&1
UNIT=1 timeout -k 10 300 python scripts/probes/search_bench.py $S > $O/search_unit.log 2>&1
cat $O/search_plain.log $O/search_unit.log | grep rows | cut -c1-250
