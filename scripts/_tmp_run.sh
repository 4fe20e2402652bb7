set -e
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py -x -q 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn -- python3 scripts/dev_encode_bench.py > gpurun_out/attn_out.txt 2>&1 || true
python scripts/pmc_summary.py stats gpurun_out/attn gpurun_out/attn_kernel_stats.csv
find gpurun_out/attn -name "*kernel_trace.csv" -delete
head -6 gpurun_out/attn_kernel_stats.csv | cut -c1-150
