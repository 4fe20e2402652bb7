set -x
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "skinny or small_batches" -s > $O/enc_tests.log 2>&1; echo "rc=$?" >> $O/enc_tests.log; tail -5 $O/enc_tests.log
timeout -k 10 300 python scripts/probes/small_encode_profile.py > $O/small_encode.log 2>&1; grep "ms per" $O/small_encode.log
CRAG_ENC_NO_SKINNY=1 timeout -k 10 300 python scripts/probes/small_encode_profile.py > $O/small_encode_noskinny.log 2>&1; grep "ms per" $O/small_encode_noskinny.log
timeout -k 10 600 python -m pytest tests/test_prefilter_gpu.py tests/test_search_gpu.py tests/test_configs_gpu.py tests/test_subnormal_bound.py tests/test_cabi.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/enc_trace -- python3 $R/scripts/probes/small_encode_profile.py > /dev/null 2>&1
cd $R
python scripts/pmc_summary.py stats $O/enc_trace $O/small_encode_kernel_stats.csv; rm -rf $O/enc_trace
head -16 $O/small_encode_kernel_stats.csv | cut -c1-150
