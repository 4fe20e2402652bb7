set -x
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "skinny or small_batches or split_k" -s > $O/enc_tests.log 2>&1; echo "rc=$?" >> $O/enc_tests.log; tail -5 $O/enc_tests.log
timeout -k 10 300 python scripts/probes/small_encode_profile.py > $O/small_encode.log 2>&1; grep "ms per" $O/small_encode.log
CRAG_ENC_NO_SKINNY_ACC=1 timeout -k 10 300 python scripts/probes/small_encode_profile.py > $O/small_encode_noacc.log 2>&1; grep "ms per" $O/small_encode_noacc.log
CRAG_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 100 --warmup 10 --rounds 2 --no-encode --no-cpu-baseline > $O/bench_n2_gloo_rehearsal.json 2> $O/bench_n2.err; echo "n2 rc=$?"
grep '^{' $O/bench_n2_gloo_rehearsal.json | cut -c1-600
