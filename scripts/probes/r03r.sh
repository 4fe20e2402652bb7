set -x
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_prefilter_gpu.py tests/test_search_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout -k 10 300 python scripts/probes/search_bench.py 1000000,64,128 100000,64,128 100000,64,10 > $O/search.log 2>&1
grep "rows" $O/search.log | cut -c1-250
