set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
CRAG_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline --encode-steps 1 2>gpurun_out/n2.err | tail -1 | cut -c1-900
for nq in 128 256; do ROWS=100000 NQ=$nq K=10 python scripts/dev_time.py; done
ROWS=100000 NQ=64 K=50 python scripts/dev_time.py
