set -x
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_prefilter_gpu.py -m gpu -x -q > $O/pf_tests.log 2>&1; echo "rc=$?" >> $O/pf_tests.log; tail -3 $O/pf_tests.log
CASES=250 SEED=3 timeout -k 10 500 python tests/stress_search.py > $O/stress.log 2>&1; tail -4 $O/stress.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_headline -- python3 $R/bench.py --no-cpu-baseline --no-encode --no-target-1m --no-fp32-rows-leg --no-overlap-leg > $R/$O/headline_line.json 2> $R/$O/stats_headline.err
cd $R
python scripts/pmc_summary.py stats $O/stats_headline $O/r03_headline_leg_kernel_stats.csv; rm -rf $O/stats_headline
grep "crag::" $O/r03_headline_leg_kernel_stats.csv | cut -c1-160
python -c "
import json; d=json.load(open('$O/headline_line.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['kernel_avg_us'], r['kernel_event_interval_us'], r['event_pair_overhead_us'])"
