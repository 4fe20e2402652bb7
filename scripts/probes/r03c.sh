set -x
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?" >> $O/bench20.err
export TMPDIR=/tmp; R=$PWD; cd /tmp
CRAG_DENSE_LIB=$R/cadence_rag_amd/csrc/libcrag_exp_base.so rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_base -- python3 $R/scripts/probes/search_bench.py 100000,64,10 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_new -- python3 $R/scripts/probes/search_bench.py 100000,64,10 > /dev/null 2>&1
cd $R
python scripts/probes/trace_gaps.py $(find $O/trace_base -name "*kernel_trace.csv" | head -1) > $O/gaps_base.log 2>&1
python scripts/probes/trace_gaps.py $(find $O/trace_new -name "*kernel_trace.csv" | head -1) > $O/gaps_new.log 2>&1
rm -rf $O/trace_base $O/trace_new
tail -5 $O/tests.log; cat $O/gaps_base.log $O/gaps_new.log; tail -2 $O/bench20.err
