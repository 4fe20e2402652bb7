set -x
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py tests/test_search_gpu.py tests/test_configs_gpu.py tests/test_subnormal_bound.py tests/test_cabi.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
python scripts/probes/search_bench.py 100000,64,10 100000,32,10 100000,1,10 1000000,64,10 100000,64,100 1000000,1,50 > $O/search.log 2>&1
cat $O/search.log | cut -c1-330
python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "small_batches" > $O/enc_tests.log 2>&1; tail -3 $O/enc_tests.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/enc_trace -- python3 $R/scripts/probes/small_encode_profile.py > $R/$O/small_encode.log 2>&1
cd $R
cat $O/small_encode.log | grep "ms per forward"
python scripts/pmc_summary.py stats $O/enc_trace $O/small_encode_kernel_stats.csv; rm -rf $O/enc_trace
head -30 $O/small_encode_kernel_stats.csv | cut -c1-200
