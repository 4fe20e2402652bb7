set -x
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_prefilter_gpu.py tests/test_search_gpu.py tests/test_configs_gpu.py tests/test_subnormal_bound.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
S="1000000,1,50 1000000,64,50 100000,64,50 100000,64,56 1000000,64,100 100000,64,100 100000,64,128 100000,64,110 1000000,64,128 100000,64,10"
timeout -k 10 300 python scripts/probes/search_bench.py $S > $O/search.log 2>&1
grep "rows" $O/search.log | cut -c1-250
CASES=200 SEED=13 timeout -k 10 300 python tests/stress_search.py > $O/stress.log 2>&1; tail -2 $O/stress.log
