# A/B of the prefilter kernel's prologue/epilogue rewrite + ablations (developer probe, run through gpurun)
set -x
O=gpurun_out/r03b; mkdir -p $O
S="100000,64,10 100000,32,10 1000000,64,10 100000,64,100"
for rep in 1 2; do
  CRAG_DENSE_LIB=$PWD/cadence_rag_amd/csrc/libcrag_exp_base.so python scripts/probes/search_bench.py $S >> $O/base.log 2>&1
  python scripts/probes/search_bench.py $S >> $O/new.log 2>&1
done
for a in 1 2 4 8 16 9 6 15 31; do
  echo "ablate=$a" >> $O/ablate.log
  CRAG_PF_ABLATE=$a python scripts/probes/search_bench.py 100000,64,10 1000000,64,10 >> $O/ablate.log 2>&1
done
python -m pytest tests/test_prefilter_gpu.py tests/test_subnormal_bound.py tests/test_search_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/scripts/probes/search_bench.py 100000,64,10 > /dev/null 2>&1
cd $R
python scripts/probes/trace_gaps.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/gaps.log 2>&1
rm -rf $O/trace
tail -3 $O/tests.log; cat $O/base.log $O/new.log $O/ablate.log $O/gaps.log
