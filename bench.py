#!/usr/bin/env python3
"""bench.py — dense-lane benchmark (contract: see DESIGN.md "Measurement").

A "step" is one pass of the hot path over one batch: QUERIES_PER_STEP fp32 query vectors
(already resident in HBM) -> exact cosine top-K over this rank's 100k x 1024 fp32 corpus shard
(BASELINE.json configs[1]) -> [N>1 only] RCCL all-gather of the per-shard top-k + on-GPU merge.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` counts the query-over-shard units all ranks processed per
second (N=1: plain queries/sec over the 100k corpus); `config.distinct_queries_per_s` is the
end-to-end rate of distinct queries answered over the whole N x 100k corpus.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROWS_PER_GPU = 100_000
DIM = 1024
TOPK = 10
QUERIES_PER_STEP = 64   # one pass of the 64-query kernel (two 32-query MFMA blocks per corpus fragment)
FP32_MFMA_PEAK_TFS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def synth(rows: int, seed: int, device) -> torch.Tensor:
    """standard normal rows, L2-normalised, fixed seed."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(rows, DIM, generator=g, dtype=torch.float32)
    x /= x.norm(dim=1, keepdim=True)
    return x.to(device)


def cpu_baseline(corpus_host: np.ndarray, queries_host: np.ndarray, gpu_ids: np.ndarray):
    """Time the CPU restatement of the reference's exact scan (oracle/, kind "port") on a bounded
    sample of the same workload, and use its result to check recall@10 of the GPU answer."""
    import oracle

    cores = os.cpu_count() or 1
    oracle.set_threads(1)
    nq1 = 8
    t0 = time.perf_counter()
    oracle.exact_topk(queries_host[:nq1], corpus_host, TOPK, mode=oracle.F32SEQ, fast=True)
    t1 = time.perf_counter() - t0
    single = nq1 / t1
    threads = min(cores, 64)
    oracle.set_threads(threads)
    nqa = min(len(queries_host), max(threads, 32))
    reps = 0
    t0 = time.perf_counter()
    while True:
        ids_f32, _, _ = oracle.exact_topk(queries_host[:nqa], corpus_host, TOPK, mode=oracle.F32SEQ, fast=True)
        reps += 1
        if time.perf_counter() - t0 > 6.0 or reps >= 20:
            break
    ta = time.perf_counter() - t0
    allc = reps * nqa / ta
    # recall@10 vs the fp64 truth oracle (eval/run_eval.py:52-55 definition)
    truth, _, _ = oracle.exact_topk(queries_host[:nqa], corpus_host, TOPK, mode=oracle.F64, fast=True)
    hits = sum(len(set(truth[i].tolist()) & set(gpu_ids[i].tolist())) for i in range(nqa))
    recall = hits / float(nqa * TOPK)
    same_order = bool(np.array_equal(truth, gpu_ids[:nqa]))
    return {
        "value": round(allc, 2), "unit": "queries/sec", "cores": threads, "kind": "port",
        "sample": f"{nqa} queries x {reps} reps over the same {len(corpus_host)}x{DIM} corpus, "
                  f"top-{TOPK}; oracle/exact_scan.c built with pgvector's float flags + OpenMP",
        "single_core_value": round(single, 2),
    }, recall, same_order


def cpu_encode_baseline(n_chunks: int = 2, tokens: int = 256):
    """The reference's CPU encode path is sentence-transformers over the HF `transformers` Qwen3 model; the
    same architecture (36 layers, random fp32 weights - no checkpoint offline) is run here through
    `transformers.Qwen3Model` on the host cores for a bounded sample, with the gateway's pooling."""
    import torch.nn.functional as F
    try:
        from transformers import Qwen3Config as HFConfig, Qwen3Model
        from cadence_rag_amd.encoder.qwen3 import Qwen3Config
        c = Qwen3Config()
        hf = HFConfig(hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                      num_hidden_layers=int(os.environ.get("CRAG_CPU_ENCODE_LAYERS", c.num_layers)),
                      num_attention_heads=c.num_heads, num_key_value_heads=c.num_kv_heads, head_dim=c.head_dim,
                      vocab_size=c.vocab_size, rms_norm_eps=c.rms_norm_eps, rope_theta=c.rope_theta,
                      max_position_embeddings=4096, tie_word_embeddings=False)
        with torch.device("meta"):
            model = Qwen3Model(hf)
        model = model.to_empty(device="cpu").float().eval()
        with torch.no_grad():
            for name, prm in model.named_parameters():
                if name.endswith("norm.weight"):
                    prm.fill_(1.0)
                else:
                    prm.uniform_(-0.02, 0.02)
            # rotary tables are buffers: rebuild them after to_empty
            for mod in model.modules():
                if hasattr(mod, "inv_freq") and hasattr(mod, "original_inv_freq"):
                    inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32) / c.head_dim))
                    mod.inv_freq = inv
                    mod.original_inv_freq = inv
            ids = torch.randint(0, c.vocab_size, (n_chunks, tokens), generator=torch.Generator().manual_seed(7))
            t0 = time.perf_counter()
            hs = model(input_ids=ids).last_hidden_state
            emb = F.normalize(hs[:, -1, : c.out_dim].float(), dim=-1)
            dt = time.perf_counter() - t0
        layers = hf.num_hidden_layers
        return {"value": round(n_chunks / dt * (layers / c.num_layers), 3), "unit": "chunks/sec",
                "cores": torch.get_num_threads(), "kind": "reference-stack (transformers Qwen3Model, fp32, CPU)",
                "sample": f"{n_chunks} chunks x {tokens} tokens, {layers} of {c.num_layers} layers timed"
                          + ("" if layers == c.num_layers else " (rate scaled to the full depth)"),
                "seconds": round(dt, 2), "finite": bool(torch.isfinite(emb).all())}
    except Exception as exc:  # the search baseline must still be reported
        return {"value": None, "unit": "chunks/sec", "error": f"{type(exc).__name__}: {exc}"}


def encode_leg(dev, rank: int, world: int, dist, steps: int):
    """chunks embedded/sec (second half of BASELINE.json's metric; configs[3] shape): batch = 256
    synthetic chunks, lengths ~N(256, 96) clipped to [8, 1024] and rescaled to mean 256, packed (no
    pad FLOPs), full Qwen3-Embedding-4B architecture with seeded random bf16 weights (no checkpoint
    is reachable offline; throughput is value-independent).  Data-parallel replicas: no collective."""
    from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder

    cfg = Qwen3Config()
    enc = Qwen3Encoder.random_init(cfg, seed=1234, device=dev)
    rng = np.random.default_rng(2024 + rank)
    n_chunks = 256
    lens = np.clip(rng.normal(256, 96, size=n_chunks).round().astype(int), 8, 1024)
    lens = (lens * (256 * n_chunks / lens.sum())).round().astype(int).clip(8, 1024)
    batch = PackedBatch.build(lens, dev)
    ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=int(lens.sum())).astype(np.int32)).to(dev)
    enc.forward_packed(ids, batch)  # warmup (GEMM autotune, allocator)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = enc.forward_packed(ids, batch)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tokens = int(lens.sum())
    ctx = float((lens.astype(float) ** 2).sum() / lens.sum())
    tflops = cfg.flops_per_token(ctx) * tokens * steps / dt / 1e12
    ok = bool(torch.isfinite(out).all().item()) and bool(torch.allclose(out.norm(dim=1), torch.ones(n_chunks, device=dev), atol=1e-3))
    del enc
    torch.cuda.empty_cache()
    return {
        "metric": "chunks embedded/sec", "value": round(world * n_chunks * steps / dt, 2), "unit": "chunks/sec",
        "tokens_per_s": round(world * tokens * steps / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 2),
        "steps": steps, "batch_chunks": n_chunks, "avg_tokens": round(tokens / n_chunks, 1), "dtype": "bf16",
        "model": "Qwen3-Embedding-4B architecture (36L, 2560h, 32q/8kv x128, 9728 ffn), seeded random weights",
        "pooling": "last token -> [:1024] -> L2 normalise", "parallelism": "replicas" if world > 1 else "1 GPU",
        "outputs_unit_norm": ok,
        # per GPU: `tflops` is this rank's own batch over the slowest rank's time
        "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(tflops / 2500.0, 4), "traffic": None,
                     "note": "per GPU; whole forward (library GEMMs + HIP ops), algorithmic FLOPs 2*P + causal attention"},
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows-per-gpu", type=int, default=ROWS_PER_GPU)
    ap.add_argument("--queries", type=int, default=QUERIES_PER_STEP)
    ap.add_argument("--topk", type=int, default=TOPK)
    ap.add_argument("--streams", type=int, default=1,
                    help="issue consecutive steps round-robin on this many HIP streams (independent query "
                         "batches may overlap on the GPU); 1 = strictly one step after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode", action="store_true", help="skip the chunks-embedded/sec leg")
    ap.add_argument("--encode-steps", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one rank per GPU; a rehearsal on a box with fewer GPUs than ranks folds ranks onto the GPUs it has
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist  # type: ignore
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CRAG_BENCH_BACKEND", "nccl")  # "gloo" only for rehearsals on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from cadence_rag_amd.dense_index import DenseIndex, ResultRecord, merge_topk_packed

    rows, nq, k = args.rows_per_gpu, args.queries, args.topk
    corpus = synth(rows, 1234 + rank, dev)
    queries = synth(max(nq, 64), 4321, dev)[:nq].contiguous()  # same queries on every rank
    ids = torch.arange(rank * rows, (rank + 1) * rows, dtype=torch.int64, device=dev)
    index = DenseIndex(DIM, capacity=rows, device=dev_index)
    index.add(corpus, ids)

    rec = ResultRecord(nq, k, dev)  # the search writes straight into the record that gets all-gathered
    out_ids, out_sc, out_ct = rec.ids, rec.scores, rec.counts
    if world > 1:
        gathered = torch.empty(world * rec.nbytes, dtype=torch.uint8, device=dev)
        f_ids = torch.empty_like(out_ids)
        f_sc = torch.empty_like(out_sc)
        f_ct = torch.empty_like(out_ct)

    stream = torch.cuda.current_stream().cuda_stream
    extra_streams = [torch.cuda.Stream(device=dev) for _ in range(max(args.streams, 1) - 1)] if world == 1 else []
    lanes = [(stream, out_ids, out_sc, out_ct)] + [
        (s.cuda_stream, torch.empty(nq, k, dtype=torch.int64, device=dev),
         torch.empty(nq, k, dtype=torch.float32, device=dev), torch.empty(nq, dtype=torch.int32, device=dev))
        for s in extra_streams]
    counter = [0]

    def step() -> None:
        if len(lanes) > 1:  # independent batches, round-robin over the streams
            st_, oi_, os_, oc_ = lanes[counter[0] % len(lanes)]
            counter[0] += 1
            index.search_async(queries, k, oi_, os_, oc_, stream=st_)
            return
        index.search_async(queries, k, out_ids, out_sc, out_ct, stream=stream)
        if world > 1:  # the path's one exchange step: ONE all-gather of 12*Q*k + 4*Q bytes per rank over xGMI
            dist.all_gather_into_tensor(gathered, rec.buf)
            merge_topk_packed(gathered, world, nq, k, f_ids, f_sc, f_ct, stream=stream)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    index.profile_enable(8)  # HIP events around every 8th scan launch of the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    n_launch, scan_ms, merge_ms = index.profile_read()
    index.profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    encode = None
    if not args.no_encode:
        encode = encode_leg(dev, rank, world, dist, args.encode_steps)

    if rank == 0:
        geo = index.scan_geometry(nq)
        scan_avg_s = scan_ms / max(n_launch, 1) / 1e3
        achieved = geo["algorithmic_bytes"] / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0
        # fp32 MFMA work of one launch: 2*Q*N*D with Q rounded up to whole 32-query MFMA blocks
        q_pad = ((nq + 31) // 32) * 32
        flops = 2.0 * q_pad * rows * DIM
        tflops = flops / scan_avg_s / 1e12 if scan_avg_s > 0 else 0.0
        # > 32 queries per pass: intensity Q/2 = 32 flop/B is past the 19.7 flop/B ridge -> matrix-pipe bound
        mfma_bound = nq > 32  # 64 queries per pass: intensity 32 flop/B, above the fp32-MFMA ridge
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
        if os.path.exists(tpath) and rows == ROWS_PER_GPU and k == TOPK:
            try:  # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/traffic.json says how)
                traffic = json.load(open(tpath))["by_queries_per_step"][str(nq)]["scan_kernel_hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": "queries/sec @ recall@10=1.0 (exact cosine top-10, 1024-d fp32 corpus)",
            "value": round(world * nq * args.steps / elapsed, 2),
            "unit": "queries/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: brute-force cosine top-{k}, {rows}x{DIM} fp32 corpus "
                            f"per GPU, {nq} queries/step resident in HBM",
                "rows_per_gpu": rows, "rows_total": rows * world, "dim": DIM, "k": k,
                "queries_per_step": nq, "streams": len(lanes),
                "parallelism": "1 GPU" if world == 1 else f"corpus sharded x{world}, all-gather top-k merge",
                "distinct_queries_per_s": round(nq * args.steps / elapsed, 2),
                "encode": "not included in this step (encoder lane reported separately when built)",
            },
            "roofline": ({
                "bound": "mfma", "achieved": round(tflops, 1), "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                "frac": round(tflops / FP32_MFMA_PEAK_TFS, 4), "traffic": traffic,
                "kernel": f"crag::scan_pipe2_kernel<{1 if k <= 32 else (2 if k <= 64 else 4)}, 2>", "kernel_avg_us": round(scan_avg_s * 1e6, 2),
                "hbm_achieved_gbs": round(achieved, 1), "hbm_frac": round(achieved / HBM_PEAK_GBS, 4),
            } if mfma_bound else {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "crag::scan_pipe_kernel" if k <= 32 else f"crag::scan_pipe2_kernel<{2 if k <= 64 else 4}, 1>",
                "kernel_avg_us": round(scan_avg_s * 1e6, 2),
                "mfma_tflops": round(tflops, 1), "mfma_frac": round(tflops / FP32_MFMA_PEAK_TFS, 4),
            }) | {
                "merge_avg_us": round(merge_ms / max(n_launch, 1) * 1e3, 2),
                "algorithmic_bytes_per_launch": geo["algorithmic_bytes"],
                "workgroups": geo["workgroups"], "launches_timed": n_launch,
            },
        }
        if encode is not None:
            line["encode"] = encode
            line["config"]["encode"] = "see top-level 'encode' (chunks embedded/sec, BASELINE configs[3] shape)"
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (rank 0's host cores)
            # rank 0's own shard result (before the cross-shard merge) vs the oracle on that shard
            base, recall, same_order = cpu_baseline(corpus.cpu().numpy(), queries.cpu().numpy(),
                                                    out_ids.cpu().numpy())
            if encode is not None:
                base["encode"] = cpu_encode_baseline()
            line["cpu_baseline"] = base
            line["config"]["recall_at_10_vs_fp64_oracle"] = recall
            line["config"]["topk_order_identical_to_oracle"] = same_order
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    index.close()


if __name__ == "__main__":
    main()
