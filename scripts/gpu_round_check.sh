set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r01/gpu_tests.log 2>&1
tail -3 $R/gpurun_out/r01/gpu_tests.log
python bench.py > $R/gpurun_out/r01/bench.json 2> $R/gpurun_out/r01/bench.err
cat $R/gpurun_out/r01/bench.json
python bench.py --queries 32 --no-encode --no-cpu-baseline > $R/gpurun_out/r01/bench_q32.json 2>> $R/gpurun_out/r01/bench.err
cat $R/gpurun_out/r01/bench_q32.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01/stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01/stats_bench.json 2>$R/gpurun_out/r01/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01/pmc_fetch64 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-encode > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01/pmc_write64 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-encode > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01/pmc_fetch32 -- python3 $R/bench.py --queries 32 --steps 100 --warmup 10 --no-cpu-baseline --no-encode > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01/pmc_write32 -- python3 $R/bench.py --queries 32 --steps 100 --warmup 10 --no-cpu-baseline --no-encode > /dev/null 2>&1
cd $R
for t in fetch64 write64 fetch32 write32; do python scripts/pmc_summary.py pmc gpurun_out/r01/pmc_$t gpurun_out/r01/pmc_${t}_summary.csv; done
python scripts/pmc_summary.py stats gpurun_out/r01/stats gpurun_out/r01/kernel_stats.csv
# raw traces are large: keep only summaries
rm -rf gpurun_out/r01/pmc_fetch64 gpurun_out/r01/pmc_write64 gpurun_out/r01/pmc_fetch32 gpurun_out/r01/pmc_write32
find gpurun_out/r01/stats -name "*kernel_trace.csv" -delete
head -12 gpurun_out/r01/kernel_stats.csv
# 1M-row single-GPU lines (north-star target: >= 70 % of the HBM roofline at 1M x 1024)
python bench.py --rows-per-gpu 1000000 --queries 32 --steps 50 --warmup 5 --no-encode --no-cpu-baseline > gpurun_out/r01/bench_1m_q32.json 2>> gpurun_out/r01/bench.err
python bench.py --rows-per-gpu 1000000 --queries 64 --steps 50 --warmup 5 --no-encode --no-cpu-baseline > gpurun_out/r01/bench_1m_q64.json 2>> gpurun_out/r01/bench.err
cat gpurun_out/r01/bench_1m_q32.json gpurun_out/r01/bench_1m_q64.json
python scripts/hybrid_bench.py > gpurun_out/r01/hybrid.json 2>> gpurun_out/r01/bench.err
cat gpurun_out/r01/hybrid.json
