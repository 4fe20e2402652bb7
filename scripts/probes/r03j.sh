set -x
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_prefilter_gpu.py tests/test_search_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
S="100000,64,10 100000,64,100 1000000,64,100 1000000,1,50 1000000,8,50"
timeout -k 10 300 python scripts/probes/search_bench.py $S > $O/search.log 2>&1
CRAG_NO_RSPLIT=1 timeout -k 10 300 python scripts/probes/search_bench.py $S > $O/search_nosplit.log 2>&1
cat $O/search.log $O/search_nosplit.log | grep rows | cut -c1-250
CRAG_SKINNY_W16=1 timeout -k 10 200 python scripts/probes/skinny_bench.py > $O/skinny_w16.log 2>&1; grep -v amdgpu $O/skinny_w16.log
