# Regenerates everything kept under profiles/ for this round (run on the GPU box through gpurun):
#   kernel-trace stats of the default bench command, FETCH_SIZE / WRITE_SIZE passes (separate runs, as the
#   microarchitecture guide prescribes) for the three search workloads of the bench line, SQ counters.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RN=${ROUND:-r04}
O=$R/gpurun_out/$RN
mkdir -p $O
# (gpurun takes 7 minutes without output for a hang: a traced bench run is silent for longer)
( while true; do sleep 45; echo "[gpu_round_check] $(date +%T) still running"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
python bench.py > $O/bench.json 2> $O/bench.err
# the driver's own invocation (few steps between synchronisations)
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench.err
# two ranks folded onto the one GPU of this box over gloo: the N > 1 code path (shards, exchange, merge, per-rank
# breakdown) end to end; its times say nothing about xGMI
CRAG_BENCH_BACKEND=gloo timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 100 --warmup 10 --rounds 2 --no-encode > $O/bench_n2_gloo_rehearsal.json 2> $O/bench_n2.err || true
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.err
# the headline leg alone (in the full command above the 1M legs launch the same kernel template)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -- python3 $R/bench.py --no-cpu-baseline --no-encode --no-target-1m --no-fp32-rows-leg --no-overlap-leg --no-other-api --no-large-k > /dev/null 2> $O/stats_headline.err
B="--steps 200 --warmup 20 --rounds 1 --no-cpu-baseline --no-encode --no-target-1m --no-fp32-rows-leg --no-overlap-leg --no-other-api --no-large-k"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_100k64 -- python3 $R/bench.py $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_1m32 -- python3 $R/bench.py $B --rows-per-gpu 1000000 --queries 32 --steps 60 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_1m64 -- python3 $R/bench.py $B --rows-per-gpu 1000000 --queries 64 --steps 60 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_100k64k100 -- python3 $R/bench.py $B --topk 100 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/pmc_sq_1m32 -- python3 $R/bench.py $B --rows-per-gpu 1000000 --queries 32 --steps 60 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2_1m32 -- python3 $R/bench.py $B --rows-per-gpu 1000000 --queries 32 --steps 60 > /dev/null 2>&1
# one 16-token query through the 36 layers, graph replays: per-kernel durations of the five-launch layer
rocprofv3 --kernel-trace --output-format csv -d $O/trace_small -- python3 $R/scripts/probes/small_encode_trace.py > $O/small_encode_trace.log 2>&1 || true
cd $R
f=$(find $O/trace_small -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then ( grep tokens $O/small_encode_trace.log; python scripts/probes/trace_gaps_enc.py $f ) > $O/${RN}_small_layer_kernel_trace.txt 2>&1 || true; fi
python scripts/pmc_summary.py stats $O/stats $O/${RN}_bench_kernel_stats.csv
python scripts/pmc_summary.py stats $O/stats_headline $O/${RN}_headline_leg_kernel_stats.csv
for w in 100k64 1m32 1m64 100k64k100; do
  for c in FETCH_SIZE WRITE_SIZE; do python scripts/pmc_summary.py pmc $O/pmc_${c}_$w $O/${RN}_pmc_${c}_$w.csv; done
done
python scripts/pmc_summary.py pmc $O/pmc_sq_1m32 $O/${RN}_prefilter_1m_q32_sq_counters.csv
python scripts/pmc_summary.py pmc $O/pmc_sq2_1m32 $O/${RN}_prefilter_1m_q32_sq_counters2.csv
python scripts/pmc_summary.py traffic $O/${RN}_pmc_FETCH_SIZE_100k64.csv $O/${RN}_pmc_WRITE_SIZE_100k64.csv 100000x64x10 $O/traffic.json
python scripts/pmc_summary.py traffic $O/${RN}_pmc_FETCH_SIZE_1m32.csv $O/${RN}_pmc_WRITE_SIZE_1m32.csv 1000000x32x10 $O/traffic.json
python scripts/pmc_summary.py traffic $O/${RN}_pmc_FETCH_SIZE_1m64.csv $O/${RN}_pmc_WRITE_SIZE_1m64.csv 1000000x64x10 $O/traffic.json
python scripts/pmc_summary.py traffic $O/${RN}_pmc_FETCH_SIZE_100k64k100.csv $O/${RN}_pmc_WRITE_SIZE_100k64k100.csv 100000x64x100 $O/traffic.json
# raw traces are large: keep only summaries
rm -rf $O/pmc_* $O/stats $O/stats_headline $O/trace_small
# bench.py prints `DETAIL {...}` (every leg) and then the compact contract line: keep both
tail -1 $O/bench.json > $O/${RN}_bench_line.json; grep '^DETAIL ' $O/bench.json | cut -c8- > $O/${RN}_bench_detail.json
tail -1 $O/bench_steps20.json > $O/${RN}_bench_line_steps20.json; grep '^DETAIL ' $O/bench_steps20.json | cut -c8- > $O/${RN}_bench_detail_steps20.json
grep '^{' $O/bench_n2_gloo_rehearsal.json > $O/${RN}_bench_n2_gloo_rehearsal.json || true
# top-50 / top-100 probes and the selection block's phase trace
python scripts/probes/pf_lags.py 2,4 > $O/${RN}_large_k.txt 2>&1 || true
python scripts/probes/fin_phase_trace.py > $O/${RN}_selection_phases.txt 2>&1 || true
cat $O/traffic.json
head -14 $O/${RN}_bench_kernel_stats.csv | cut -c1-160
