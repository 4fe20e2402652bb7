#!/usr/bin/env python3
"""bench.py — dense-lane benchmark (contract: see DESIGN.md "Measurement").

A "step" is one pass of the hot path over one batch: QUERIES_PER_STEP fp32 query vectors (already resident in
HBM) -> exact cosine top-K over this rank's fp32 corpus shard -> [N > 1 only] the path's one exchange step, an
RCCL all-gather of the per-shard top-k + the on-GPU merge.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json):
  N = 1   configs[1]: 100 000 x 1024 corpus, 64 queries per step, top-10.  The same line carries `target_1m`
          (1M x 1024 on one GPU, 32 and 64 queries per step: the north-star >= 70 % HBM target) and `encode`
          (configs[3] shape).
  N > 1   configs[2]: ONE 1M x 1024 corpus sharded over the N ranks (contiguous row shards), every rank searches
          the same 64 queries, one all-gather, merge.  Total work is fixed: "scaling": "strong"; the 1-GPU point
          of that curve is `target_1m.q64` of the N = 1 line.  `--mode weak` instead keeps 100 000 rows per rank.
`value` is always DISTINCT queries answered per second by the whole job (never multiplied by the rank count).
Rank 0 prints a `DETAIL {...}` line with every leg, then LAST the ONE compact JSON line (< 4 KB) the driver parses.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROWS_CONFIG1 = 100_000
ROWS_CONFIG2 = 1_000_000
DIM = 1024
TOPK = 10
QUERIES_PER_STEP = 64
FP32_MFMA_PEAK_TFS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0       # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
ROOT = os.path.dirname(os.path.abspath(__file__))


SYNTH_CHUNK = 125_000


def synth_rows(lo: int, hi: int, seed: int, device) -> torch.Tensor:
    """Rows [lo, hi) of the synthetic matrix of SURVEY.md 8(d): standard normal rows, L2-normalised, fixed seed.
    The matrix is DEFINED in chunks of 125 000 rows, chunk i from the device generator seeded seed*1000 + i, so a
    row's value does not depend on how many rows are asked for or on which rank generates it (a 1M x 1024
    corpus is 4.1 GB: generated on the device, chunk by chunk)."""
    out = torch.empty(hi - lo, DIM, dtype=torch.float32, device=device)
    for ci in range(lo // SYNTH_CHUNK, (hi + SYNTH_CHUNK - 1) // SYNTH_CHUNK):
        g = torch.Generator(device=device).manual_seed(seed * 1000 + ci)
        blk = torch.randn(SYNTH_CHUNK, DIM, generator=g, device=device, dtype=torch.float32)
        blk /= blk.norm(dim=1, keepdim=True)
        a, b = max(lo, ci * SYNTH_CHUNK), min(hi, (ci + 1) * SYNTH_CHUNK)
        out[a - lo:b - lo] = blk[a - ci * SYNTH_CHUNK:b - ci * SYNTH_CHUNK]
        del blk
    return out


def synth(rows: int, seed: int, device) -> torch.Tensor:
    return synth_rows(0, rows, seed, device)


def timed_rounds(step, fence, steps: int, rounds: int):
    """`rounds` repetitions of EXACTLY `steps` steps, each bracketed by fence() on both sides."""
    times = []
    for _ in range(rounds):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        times.append(time.perf_counter() - t0)
    return times


def search_leg(index, queries, k, steps, warmup, rounds, step_extra=None, fence_extra=None, outs=None,
               prewarm_s=0.3, pipelined=False):
    """Warm up, then time `rounds` x `steps` searches of `queries` (device tensor) on torch's current stream.
    Returns timing + live HIP-event kernel times + the prefilter path's candidate statistics.
    pipelined: the steps go through crag_index_search_pipelined (the throughput form for a run of independent
    searches from one stream: consecutive searches rotate over three streams of the index's own) with ONE
    crag_index_join in front of every fence -- the queries are resident and the outputs are read behind the fence."""
    dev = queries.device
    nq = int(queries.shape[0])
    if outs is None:
        outs = (torch.empty(nq, k, dtype=torch.int64, device=dev), torch.empty(nq, k, dtype=torch.float32, device=dev),
                torch.empty(nq, dtype=torch.int32, device=dev))
    oi, osc, oc = outs
    stream = torch.cuda.current_stream().cuda_stream

    if pipelined:   # an output set per internal stream (up to 4): consecutive searches run on several streams at once
        alts = [(oi, osc, oc)] + [(torch.empty_like(oi), torch.empty_like(osc), torch.empty_like(oc)) for _ in range(3)]
        alt = alts[1]
        flip = [0]

    def step():
        if pipelined:
            flip[0] = (flip[0] + 1) % 4
            index.search_pipelined(queries, k, *alts[flip[0]], stream=stream, inputs_ready=True)
        else:
            index.search_async(queries, k, oi, osc, oc, stream=stream)
        if step_extra is not None:
            step_extra()

    def fence():
        if pipelined:
            index.join(stream)
        if fence_extra is not None:
            fence_extra()
        torch.cuda.synchronize()

    if step_extra is None:
        t_end = time.perf_counter() + prewarm_s  # clocks ramp over ~0.1 s; untimed
        while time.perf_counter() < t_end:
            for _ in range(16):
                step()
            torch.cuda.synchronize()
    else:
        # N > 1: a step contains a collective, so every rank must run the SAME number of steps -- a clock-bounded
        # loop lets one rank issue 16 all-gathers more than its peers and the job hangs at its end
        for _ in range(int(200 * prewarm_s / 0.3) + 16):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    fence()
    index.prefilter_stats()
    # Live HIP-event samples on the launch stream, SPARSE: a sampled search carries five extra event packets = 19 us
    # (scripts/probes/round_fixed_cost.py: a round of 20 searches takes 976 us without sampling, 1024 us with every
    # 8th search sampled; at the driver's --steps 20, sampling EVERY search (round 2) stretched the timed step from
    # 57 to 76 us).  At most every 20th search of the timed region is sampled (the one in the middle of its window of
    # 20: one per round of the driver's form); the same workload then continues untimed until 64 (short runs: 24) samples exist.
    total = steps * rounds                     # the library samples the searches c with c % every == every // 2
    want = 64 if total >= 1000 else 24         # a short run (the driver's --steps 20) continues for 24 samples only
    every = max(20, total // want)
    index.profile_enable(every)
    times = timed_rounds(step, fence, steps, rounds)
    in_region = (total - every // 2 - 1) // every + 1 if total > every // 2 else 0
    for _ in range(max(0, want - in_region) * every):
        step()
    fence()
    n_launch, scan_ms, rest_ms, pair_ms = index.profile_read_ex()
    index.profile_enable(0)
    stats = index.prefilter_stats()
    per = max(n_launch, 1)
    return {"times": times, "n_launch": n_launch, "samples_in_timed_region": min(in_region, n_launch),
            "scan_raw_us": scan_ms / per * 1e3, "event_pair_us": pair_ms / per * 1e3,
            # An event-to-event interval around ONE kernel = dispatch + kernel + the trailing event's completion.  Two
            # events recorded back to back (nothing between) are `event_pair_us` apart (5.5 us on MI355X / ROCm 7.2):
            # two serialised event completions.  Half of it is taken as the one completion inside the interval;
            # measured against rocprofv3's own begin/end timestamps of the same kernel in the same command this
            # estimate agrees within 1 % (100 000 x 64: 38.6 us interval, 5.5 us pair -> 35.8; rocprofv3 35.4),
            # the raw interval is 9 % long and interval - pair 7 % short.  All three numbers are printed.
            "scan_us": max(scan_ms - 0.5 * pair_ms, 0.0) / per * 1e3,
            "rest_us": rest_ms / per * 1e3, "stats": stats, "kernel": index.last_scan_kernel(),
            "outputs_identical_on_both_streams": (bool(torch.equal(oi, alt[0]) and torch.equal(oc, alt[2]))
                                                  if pipelined else None),
            "row_bytes": (index.prefilter_row_bytes() if "prefilter" in index.last_scan_kernel() else DIM * 4),
            "out": (oi, osc, oc)}


def overlap_leg(index, queries, k, n_streams, steps):
    """The same steps issued round-robin on `n_streams` HIP streams: consecutive steps are independent query
    batches, so the scan of one may run beside the rescoring of another and the memory system never idles between
    scans.  Throughput only: per-kernel times overlap, the roofline is quoted from the one-stream run."""
    dev = queries.device
    nq = int(queries.shape[0])
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    outs = [(torch.empty(nq, k, dtype=torch.int64, device=dev), torch.empty(nq, k, dtype=torch.float32, device=dev),
             torch.empty(nq, dtype=torch.int32, device=dev)) for _ in range(n_streams)]
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            s = i % n_streams
            index.search_async(queries, k, *outs[s], stream=streams[s].cuda_stream)
        torch.cuda.synchronize()

    run(max(steps // 10, n_streams))
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    same = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
    return {"value": round(nq * steps / dt, 2), "unit": "queries/sec", "ms_per_step": round(dt / steps * 1e3, 5),
            "steps": steps, "results_identical_across_streams": bool(same)}


def roofline(rows, nq, k, leg, traffic_doc):
    """algorithmic bytes per launch = N * (row bytes the scan must stream) + Q*D*4 + Q*k*12, / the
    scan kernel's live event time.  Row bytes: D*4 for a scan of the fp32 rows; D*2 when the index keeps the fp16
    mirror of the unit rows and the prefilter scan streams that instead (8(d): a prefilter's bytes are declared
    separately -- the fp32 rows are then read only for the rescored candidates, reported below)."""
    row_bytes = leg.get("row_bytes") or DIM * 4
    alg = rows * row_bytes + nq * DIM * 4 + nq * k * 12
    scan_s = leg["scan_us"] * 1e-6
    gbs = alg / scan_s / 1e9 if scan_s > 0 else 0.0
    q_pad = ((nq + 31) // 32) * 32
    per = max(leg["stats"]["searches"], 1)
    prefilter = "prefilter" in leg["kernel"]
    mfma_tfs = 2.0 * q_pad * rows * DIM / scan_s / 1e12 if scan_s > 0 else 0.0
    out = {
        "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
        "kernel": leg["kernel"].replace("crag::", ""), "kernel_avg_us": round(leg["scan_us"], 2),
        # how kernel_avg_us is derived from the three event numbers: DESIGN.md 5
        "kernel_event_interval_us": round(leg.get("scan_raw_us", leg["scan_us"]), 2),
        "event_pair_overhead_us": round(leg.get("event_pair_us", 0.0), 2),
        "other_kernels_avg_us": round(leg["rest_us"], 2), "launches_timed": leg["n_launch"],
        "samples_in_timed_region": leg.get("samples_in_timed_region"),
        "algorithmic_bytes_per_launch": alg, "row_bytes_streamed": row_bytes,
        "cache_assisted": bool(rows * row_bytes < 256 * 2**20),   # streamed bytes fit the 256 MiB Infinity Cache
        "matrix_pipe_tflops": round(mfma_tfs, 1), "matrix_pipe": "fp16 MFMA" if prefilter else "fp32 MFMA",
    }
    if not prefilter:
        out["matrix_pipe_frac"] = round(mfma_tfs / FP32_MFMA_PEAK_TFS, 3)
    if row_bytes != DIM * 4:   # N*D*4 / the same time: what a scan of the fp32 rows would need; NOT a roofline quantity
        out["fp32_rows_equivalent_GBs"] = round(rows * DIM * 4 / scan_s / 1e9, 1) if scan_s > 0 else 0.0
    if prefilter:  # declared separately: rows re-read for the exact fp32 score, 4 KiB each
        out["rescored_rows_per_launch"] = round(leg["stats"]["rescored_rows"] / per, 1)
        out["rescored_bytes_per_launch"] = int(leg["stats"]["rescored_rows"] / per * DIM * 4)
        out["candidates_per_launch"] = round(leg["stats"]["candidates"] / per, 1)
    key = f"{rows}x{nq}x{k}"
    if traffic_doc and key in traffic_doc.get("by_workload", {}):
        t = traffic_doc["by_workload"][key]
        if t.get("kernel", "").split("(")[0].replace("void ", "").replace("crag::", "") == out["kernel"]:
            out["traffic"] = t["hbm_bytes_per_launch"]   # separate rocprofv3 --pmc passes, not this run
            out["traffic_source"] = "profiles/traffic.json"
    return out


def fp32_rows_leg(corpus, ids, queries, k, dev_index, steps, ref_out, traffic_doc):
    """The same search over an index WITHOUT the fp16 mirror (CRAG_NO_FP16_MIRROR=1: the prefilter scan streams the
    4 KiB fp32 rows and rounds them in registers): SURVEY.md 8(d)'s literal accounting, N*D*4 bytes per launch.
    Results must equal the mirror scan's bit for bit."""
    from cadence_rag_amd.dense_index import DenseIndex
    os.environ["CRAG_NO_FP16_MIRROR"] = "1"  # read once, when the index is created
    try:
        index = DenseIndex(DIM, capacity=int(corpus.shape[0]), device=dev_index)
    finally:
        del os.environ["CRAG_NO_FP16_MIRROR"]
    try:
        index.add(corpus, ids)
        nq = int(queries.shape[0])
        leg = search_leg(index, queries, k, steps, 20, 3, prewarm_s=0.1)
        same = all(bool(torch.equal(a, b)) for a, b in zip(leg["out"], ref_out))
        return {"value": round(nq * steps / leg["times"][0], 2), "unit": "queries/sec",
                "ms_per_step": round(leg["times"][0] / steps * 1e3, 5), "steps": steps,
                "results_identical_to_mirror_scan": same,
                "roofline": roofline(int(corpus.shape[0]), nq, k, leg, traffic_doc)}
    finally:
        index.close()


def recall_check(corpus_host, queries_host, gpu_ids, k, sample):
    """recall@10 (eval/run_eval.py:52-55) and order identity of the GPU answer vs the fp64 oracle on `sample`."""
    import oracle
    oracle.set_threads(min(os.cpu_count() or 1, 64))
    truth, _, _ = oracle.exact_topk(queries_host[sample], corpus_host, k, mode=oracle.F64, fast=True)
    hits = sum(len(set(truth[i, :10].tolist()) & set(gpu_ids[s, :10].tolist())) for i, s in enumerate(sample))
    return hits / float(len(sample) * min(k, 10)), bool(np.array_equal(truth, gpu_ids[sample]))


def cpu_baseline(corpus_host: np.ndarray, queries_host: np.ndarray):
    """The CPU restatement of the reference's exact scan (oracle/, kind "port") timed on a bounded sample of the
    same workload on this box's host cores."""
    import oracle
    cores = os.cpu_count() or 1
    oracle.set_threads(1)
    nq1 = 8
    t0 = time.perf_counter()
    oracle.exact_topk(queries_host[:nq1], corpus_host, TOPK, mode=oracle.F32SEQ, fast=True)
    single = nq1 / (time.perf_counter() - t0)
    threads = min(cores, 64)
    oracle.set_threads(threads)
    nqa = min(len(queries_host), max(threads, 32))
    reps = 0
    t0 = time.perf_counter()
    while True:
        oracle.exact_topk(queries_host[:nqa], corpus_host, TOPK, mode=oracle.F32SEQ, fast=True)
        reps += 1
        if time.perf_counter() - t0 > 6.0 or reps >= 20:
            break
    allc = reps * nqa / (time.perf_counter() - t0)
    return {
        "value": round(allc, 2), "unit": "queries/sec", "cores": threads, "kind": "port",
        "sample": f"{nqa} queries x {reps} reps over the same {len(corpus_host)}x{DIM} corpus, "
                  f"top-{TOPK}; oracle/exact_scan.c built with pgvector's float flags + OpenMP",
        "single_core_value": round(single, 2),
    }


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
            for mod in model.modules():  # rotary tables are buffers: rebuild them after to_empty
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


def encode_leg(dev, rank: int, world: int, dist, steps: int, enc=None):
    """chunks embedded/sec (second half of BASELINE.json's metric; configs[3] shape): batch = 256 synthetic
    chunks, lengths ~N(256, 96) clipped to [8, 1024] and rescaled to mean 256, packed (no pad FLOPs), full
    Qwen3-Embedding-4B architecture with seeded random bf16 weights (no checkpoint is reachable offline;
    throughput is value-independent).  Data-parallel replicas: no collective.  Two rates: `value` times
    forward_packed on pre-tokenised device ids (the kernel-level rate); `backfill_path` drives the real
    run_embedding_backfill loop (texts -> tokeniser -> packing -> forward -> store -> HBM index sink)."""
    from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder

    cfg = Qwen3Config()
    if enc is None:
        enc = Qwen3Encoder.random_init(cfg, seed=1234, device=dev)
    rng = np.random.default_rng(2024 + rank)
    n_chunks = 256
    lens = np.clip(rng.normal(256, 96, size=n_chunks).round().astype(int), 8, 1024)
    lens = (lens * (256 * n_chunks / lens.sum())).round().astype(int).clip(8, 1024)
    batch = PackedBatch.build(lens, dev)
    ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=int(lens.sum())).astype(np.int32)).to(dev)
    for _ in range(2):
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
    tflops = cfg.flops_per_token(ctx) * tokens * steps / dt / 1e12   # tokens/s x FLOPs per token of the model
    # executed: minus the last layer's o / MLP projections of the rows that are not pooled (last-token pooling)
    tflops_exec = (cfg.flops_per_token(ctx) * tokens - cfg.flops_skipped_in_last_layer(tokens, n_chunks)) * steps / dt / 1e12
    ok = bool(torch.isfinite(out).all().item()) and bool(torch.allclose(out.norm(dim=1), torch.ones(n_chunks, device=dev), atol=1e-3))
    res = {
        "metric": "chunks embedded/sec", "value": round(world * n_chunks * steps / dt, 2), "unit": "chunks/sec",
        "tokens_per_s": round(world * tokens * steps / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 2),
        "steps": steps, "batch_chunks": n_chunks, "avg_tokens": round(tokens / n_chunks, 1), "dtype": "bf16",
        "model": "Qwen3-Embedding-4B architecture (36L, 2560h, 32q/8kv x128, 9728 ffn), seeded random weights",
        "pooling": "last token -> [:1024] -> L2 normalise", "parallelism": "replicas" if world > 1 else "1 GPU",
        "outputs_unit_norm": ok,
        # per GPU: `tflops` is this rank's own batch over the slowest rank's time
        "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(tflops / 2500.0, 4), "traffic": None,
                     # per GPU; achieved = SURVEY 8(d)'s algorithmic FLOPs (2*P + causal attention per token) / time;
                     # executed_* leaves out the last layer's o / MLP projections of the non-pooled rows
                     "executed_tflops": round(tflops_exec, 1), "executed_frac": round(tflops_exec / 2500.0, 4)},
    }
    if rank == 0:
        try:
            res["backfill_path"] = backfill_path_leg(enc, cfg, dev)
        except Exception as exc:  # the kernel-level rate above must still be reported
            res["backfill_path"] = {"error": f"{type(exc).__name__}: {exc}"}
    del enc
    torch.cuda.empty_cache()
    return res


def backfill_path_leg(enc, cfg, dev, n_rows: int = 1536, batch_size: int = 256):
    """run_embedding_backfill end to end (reference entry point embedding_pipeline.py:241) over synthetic texts of
    ~256 tokens: fetch -> tokenise -> pack -> forward -> store -> DenseIndex sink in HBM, in the two forms the
    store can take the vectors in: host lists (the reference's List[List[float]] contract) and device-resident."""
    from uuid import UUID
    from cadence_rag_amd import embedding_pipeline as ep, embeddings
    from cadence_rag_amd.config import settings
    from cadence_rag_amd.dense_index import DenseIndex
    from cadence_rag_amd.encoder.qwen3 import ByteTokenizer
    enc.tokenizer = ByteTokenizer(eos_id=256)
    rng = np.random.default_rng(99)
    lens = np.clip(rng.normal(255, 96, size=n_rows).round().astype(int), 8, 1000)
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz      ", dtype=np.uint8)
    texts = ["x" + bytes(alphabet[rng.integers(0, len(alphabet), size=n - 1)]).decode() for n in lens]
    old = (settings.embeddings_base_url, settings.embeddings_dim)
    settings.embeddings_base_url, settings.embeddings_dim = "native", cfg.out_dim
    embeddings.set_encoder(enc)
    out = {"rows": n_rows, "batch_size": batch_size, "avg_tokens": round(float(lens.mean()) + 1, 1),
           "tokenizer": "ByteTokenizer stand-in (utf-8 bytes; no Qwen tokenizer files offline)"}
    try:
        for mode in ("host_lists", "device_resident"):
            tables = {"chunks": {i: {"call_id": UUID(int=1 + i % 7), "text": texts[i], "embedding": None}
                                 for i in range(n_rows)}, "artifact_chunks": {}}
            with DenseIndex(cfg.out_dim, capacity=n_rows) as sink:
                store_cls = ep.InMemoryStore if mode == "host_lists" else ep.DeviceSinkStore
                ep.set_store(store_cls(tables, sinks={"chunks": sink}))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                summary = ep.run_embedding_backfill(batch_size=batch_size)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                assert summary.rows_updated == n_rows and len(sink) == n_rows
            out[mode] = {"chunks_per_s": round(n_rows / dt, 1), "seconds": round(dt, 3)}
    finally:
        ep.set_store(None)
        embeddings.set_encoder(None)
        settings.embeddings_base_url, settings.embeddings_dim = old
    return out


def hybrid_leg(big_index, rows: int, dev, k_dense: int = 100, k_tech: int = 50, nq: int = 64, steps: int = 30):
    """BASELINE configs[4]: hybrid /retrieve candidates on the GPU — dense top-100 (exact cosine) + the exact-token
    lane top-50 (retrieve.py:183-242) + BM25 ranks as GIVEN inputs (pg_search's arithmetic is not in the reference
    repository) fused by reciprocal rank (retrieve.py:245-260), batch = 64 queries over the 1M-chunk corpus, everything
    stream-ordered on the device.  A step = one HybridSearcher.search."""
    from cadence_rag_amd.fusion import HybridSearcher, TechTokenIndex, rrf_fuse
    rng = np.random.default_rng(0)
    vocab = np.array([f"TOK-{i}" for i in range(2000)])
    n_tok = rng.integers(0, 4, size=rows)
    flat = vocab[rng.integers(0, 2000, size=int(n_tok.sum()))].tolist()
    row_tokens, o = [], 0
    for n in n_tok.tolist():
        row_tokens.append(flat[o:o + n])
        o += n
    started = np.datetime64("2026-01-01", "us") + rng.integers(0, 365, size=rows).astype("timedelta64[D]")
    tech = TechTokenIndex(row_tokens, np.arange(rows), started, dev, verify=False)
    q = synth(nq, 4321, dev)
    qtoks = [vocab[rng.integers(0, 2000, size=3)].tolist() for _ in range(nq)]
    bm25_ids = torch.from_numpy(rng.integers(0, rows, size=(nq, 50))).to(dev)
    bm25_ct = torch.full((nq,), 50, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    hs = HybridSearcher(big_index, tech, dense_k=k_dense, tech_k=k_tech)

    def step():
        return hs.search(q, qtoks, (bm25_ids, bm25_ct), out_k=k_dense + k_tech + 50, stream=st)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # the same step with the lanes in series on the caller's stream (HybridSearcher(overlap_lanes=False); the default
    # forks the token lane onto a side stream behind the scan's launch and joins it in front of the fusion)
    hs_side = HybridSearcher(big_index, tech, dense_k=k_dense, tech_k=k_tech, overlap_lanes=False)
    side_step = lambda: hs_side.search(q, qtoks, (bm25_ids, bm25_ct), out_k=k_dense + k_tech + 50, stream=st)  # noqa: E731
    for _ in range(30):   # (the first ~30 steps of this leg run 5-8 % slower than its steady state)
        out = step()
        out_side = side_step()
    rounds = [(timed(step, steps), timed(side_step, steps)) for _ in range(3)]   # alternating: same conditions
    dt = statistics.median(r[0] for r in rounds)
    dt_side = statistics.median(r[1] for r in rounds)
    same = all(bool(torch.equal(out[key], out_side[key])) for key in ("ids", "scores", "lanes", "counts"))
    d_ids = torch.empty(nq, k_dense, dtype=torch.int64, device=dev)
    d_sc = torch.empty(nq, k_dense, dtype=torch.float32, device=dev)
    d_ct = torch.empty(nq, dtype=torch.int32, device=dev)
    dense_leg = search_leg(big_index, q, k_dense, 100, 10, 1, outs=(d_ids, d_sc, d_ct), prewarm_s=0.05)
    t_ids, t_ct = tech.search(qtoks, k_tech, stream=st)
    split = {"dense_top%d_us" % k_dense: round(dense_leg["times"][0] / 100 * 1e6, 1),
             "token_lane_top%d_us" % k_tech: round(timed(lambda: tech.search(qtoks, k_tech, stream=st), 20) * 1e6, 1),
             "rrf_fuse_us": round(timed(lambda: rrf_fuse([(bm25_ids, bm25_ct), (t_ids, t_ct), (d_ids, d_ct)],
                                                         out_k=k_dense + k_tech + 50, stream=st), 20) * 1e6, 1)}
    # property checks on the fused output (the parity test at this size is tests/test_configs_gpu.py)
    cnt = out["counts"].cpu().numpy()
    ok = bool((cnt >= k_dense).all()) and bool(torch.equal(out["dense_ids"], d_ids))
    return {"workload": f"BASELINE configs[4]: hybrid retrieve, {rows} chunks, batch {nq}: dense top-{k_dense} + "
                        f"exact-token lane top-{k_tech} + given BM25 ranks (50) -> RRF on the GPU",
            "ms_per_step": round(dt * 1e3, 4), "value": round(nq / dt, 1), "unit": "queries/sec", "steps": steps,
            # median of 3 rounds of `steps` steps, alternating with the in-series variant, after 30 warm steps
            "ms_per_step_lanes_in_series": round(dt_side * 1e3, 4), "results_identical_in_series": same,
            "split": split, "fused_counts_min": int(cnt.min()), "self_check_ok": ok,
            "dense_roofline": roofline(rows, nq, k_dense, dense_leg, None)}


def query_path_leg(enc, cfg, big_index, big, dev, dev_index, query_tokens: int = 16):
    """The reference's own operating point (/root/reference/app/retrieve.py:18-19,427,472-487): ONE request = embed
    ONE query string, then the dense top-50 over `chunks` and the dense top-10 over `artifact_chunks`.  Here: the
    36-layer encoder on a packed batch of nq short queries (nq = 1, 8, 64; `query_tokens` tokens each), the 1M-row
    chunks index (k = 50) and a 100 000-row artifact index (k = 10), all stream-ordered.  Reported per nq: encode
    latency with its roofline (at small batches the forward is a WEIGHT STREAM: every layer's bf16 weights are read
    once per forward), both searches with their HBM rooflines, the request latency (one synchronisation per request,
    what a /retrieve caller waits for) and the pipelined request rate."""
    from cadence_rag_amd.dense_index import DenseIndex
    n_art = 100_000
    art = DenseIndex(DIM, capacity=n_art, device=dev_index)
    art.add(big[:n_art])
    weight_bytes = 2 * cfg.num_layers * (cfg.hidden_size * (cfg.q_size + 2 * cfg.kv_size) + cfg.q_size * cfg.hidden_size
                                         + 3 * cfg.hidden_size * cfg.intermediate_size)
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(5)
    out = {"workload": "reference operating point: embed nq short queries -> dense top-50 over 1M chunks + dense "
                       "top-10 over 100 000 artifact chunks (retrieve.py:18-19,427,472-487)",
           "query_tokens": query_tokens, "encoder_weight_bytes": weight_bytes}

    def lat(fn, n):  # mean latency of n requests, each synchronised (nothing in flight when it starts)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def rate(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    try:
        for nq in (1, 8, 64):
            token_lists = [rng.integers(0, cfg.vocab_size, size=query_tokens).tolist() for _ in range(nq)]
            c_out = (torch.empty(nq, 50, dtype=torch.int64, device=dev), torch.empty(nq, 50, dtype=torch.float32, device=dev),
                     torch.empty(nq, dtype=torch.int32, device=dev))
            a_out = (torch.empty(nq, 10, dtype=torch.int64, device=dev), torch.empty(nq, 10, dtype=torch.float32, device=dev),
                     torch.empty(nq, dtype=torch.int32, device=dev))

            def encode():   # the product path from token ids on: host packing, H2D, forward (one graph replay)
                return enc.embed_token_lists(token_lists)

            os.environ["CRAG_ENC_NO_GRAPH"] = os.environ["CRAG_ENC_NO_SKINNY"] = "1"   # round 2's only path: eager
            try:                                                                       # launches, library GEMMs
                eager_lat = lat(encode, 8)
            finally:
                del os.environ["CRAG_ENC_NO_GRAPH"], os.environ["CRAG_ENC_NO_SKINNY"]
            qv = encode()

            def request():
                v = encode()
                big_index.search_async(v, 50, *c_out, stream=st)
                art.search_async(v, 10, *a_out, stream=st)

            for _ in range(3):
                request()
            torch.cuda.synchronize()
            e_lat = lat(encode, 30)
            chunks_leg = search_leg(big_index, qv, 50, 60, 5, 1, outs=c_out, prewarm_s=0.05)
            art_leg = search_leg(art, qv, 10, 100, 5, 1, outs=a_out, prewarm_s=0.05)
            r_lat = lat(request, 30)
            r_rate = rate(request, 60)
            out[f"nq{nq}"] = {
                "encode_ms": round(e_lat * 1e3, 4),
                "encode_ms_eager_launches": round(eager_lat * 1e3, 4),
                "encode_roofline": {"bound": "hbm", "achieved": round(weight_bytes / e_lat / 1e9, 1), "peak": HBM_PEAK_GBS,
                                    # algorithmic bytes = the 36 layers' bf16 weights, read once per forward
                                    "unit": "GB/s", "frac": round(weight_bytes / e_lat / 1e9 / HBM_PEAK_GBS, 4)},
                "search_chunks_k50": {"ms_per_step": round(chunks_leg["times"][0] / 60 * 1e3, 4),
                                      "roofline": roofline(ROWS_CONFIG2, nq, 50, chunks_leg, None)},
                "search_artifacts_k10": {"ms_per_step": round(art_leg["times"][0] / 100 * 1e3, 4),
                                         "roofline": roofline(n_art, nq, 10, art_leg, None)},
                "request_latency_ms": round(r_lat * 1e3, 4),
                "requests_per_s_pipelined": round(1.0 / r_rate, 1),
                "queries_per_s_pipelined": round(nq / r_rate, 1),
            }
        # encode latency alone over the shapes between one short query and the gateway's full batch (RUNBOOK:304,331-334):
        # one query of 17..32 tokens, then 2 / 4 / 6 queries of `query_tokens` tokens (32 / 64 / 96 token rows)
        sweep = {}
        for nq, ntok in ((1, 2 * query_tokens), (2, query_tokens), (4, query_tokens), (6, query_tokens)):
            tls = [rng.integers(0, cfg.vocab_size, size=ntok).tolist() for _ in range(nq)]
            enc.embed_token_lists(tls)
            ms = lat(lambda: enc.embed_token_lists(tls), 20) * 1e3
            sweep[f"{nq}x{ntok}"] = {"encode_ms": round(ms, 4), "frac_of_weight_stream": round(weight_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out["encode_latency_sweep"] = sweep
    finally:
        art.close()
    return out


COMPACT_LIMIT = 4096   # the driver keeps 8 KB of stdout: the LAST line must fit with room to spare


def _get(d, *path, default=None):
    for key in path:
        if not isinstance(d, dict) or key not in d:
            return default
        d = d[key]
    return d


def compact_line(full: dict) -> dict:
    """The ONE line the driver parses (the bench contract's keys, `roofline`, `cpu_baseline`, and one number per
    secondary leg).  Everything else is in the DETAIL line printed before it and in gpurun_out/bench_detail.json."""
    cfg = full.get("config", {})
    roof = full.get("roofline", {})
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": cfg.get("workload"), "rows": cfg.get("rows_total"), "k": cfg.get("k"),
                     "queries_per_step": cfg.get("queries_per_step"), "parallelism": cfg.get("parallelism"),
                     "arithmetic": cfg.get("arithmetic"),
                     "recall": cfg.get("recall_at_10_vs_fp64_oracle"),
                     "order_identical": cfg.get("topk_order_identical_to_oracle"),
                     "ms_per_step_median": cfg.get("ms_per_step_median")}
    for key in ("api",):
        if key in cfg:
            out["config"][key] = cfg[key]
    if cfg.get("per_rank_step_breakdown"):
        rows_ = cfg["per_rank_step_breakdown"]
        out["config"]["per_rank_us_max"] = {f: max(r.get(f, 0) for r in rows_ if r) for f in
                                            ("search_us", "exchange_us", "merge_us", "scan_kernel_us")}
        out["config"]["speedup_vs_same_job_on_one_gpu"] = cfg.get("speedup_vs_same_job_on_one_gpu")
        out["config"]["scaling_curve_measured"] = cfg.get("scaling_curve_measured")
    out["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                "kernel_avg_us", "cache_assisted", "samples_in_timed_region",
                                                "launches_timed", "algorithmic_bytes_per_launch")}
    base = full.get("cpu_baseline")
    if base is not None:
        out["cpu_baseline"] = {k: base.get(k) for k in ("value", "unit", "cores", "kind", "sample", "single_core_value")}
        out["cpu_baseline"]["encode_chunks_per_s"] = _get(base, "encode", "value")
    summ = {
        "streams2_q_per_s": _get(cfg, "steps_overlapped_on_streams", "2", "value"),
        "pipelined_api_q_per_s": _get(full, "pipelined_api", "value"),
        "in_order_api_q_per_s": _get(full, "in_order_api", "value"),
        "fp32_rows_scan_frac": _get(full, "fp32_rows_scan", "roofline", "frac"),
        "target_1m_q32_frac": _get(full, "target_1m", "q32", "roofline", "frac"),
        "target_1m_q64_frac": _get(full, "target_1m", "q64", "roofline", "frac"),
        "target_1m_q64_kernel_us": _get(full, "target_1m", "q64", "roofline", "kernel_avg_us"),
        "target_1m_q64_ms": _get(full, "target_1m", "q64", "ms_per_step"),
        "target_1m_q64_q_per_s": _get(full, "target_1m", "q64", "value"),
        "target_1m_q64_recall": _get(full, "target_1m", "q64", "recall_at_10_vs_fp64_oracle"),
        "target_1m_q64_fp32_rows_frac": _get(full, "target_1m", "q64", "fp32_rows_scan", "roofline", "frac"),
        "hybrid_ms": _get(full, "hybrid", "ms_per_step"),
        "hybrid_dense_frac": _get(full, "hybrid", "dense_roofline", "frac"),
        "hybrid_token_lane_us": _get(full, "hybrid", "split", "token_lane_top50_us"),
        "k100_100k_ms": _get(full, "large_k", "100000x64x100", "ms_per_step"),
        "k100_100k_frac": _get(full, "large_k", "100000x64x100", "roofline", "frac"),
        "k50_100k_ms": _get(full, "large_k", "100000x64x50", "ms_per_step"),
        "k100_1m_ms": _get(full, "large_k", "1000000x64x100", "ms_per_step"),
        "k100_1m_frac": _get(full, "large_k", "1000000x64x100", "roofline", "frac"),
        "nq1_encode_ms": _get(full, "query_path", "nq1", "encode_ms"),
        "nq1_encode_frac": _get(full, "query_path", "nq1", "encode_roofline", "frac"),
        "nq1_request_ms": _get(full, "query_path", "nq1", "request_latency_ms"),
        "q1x32_encode_ms": _get(full, "query_path", "encode_latency_sweep", "1x32", "encode_ms"),
        "nq4_encode_ms": _get(full, "query_path", "encode_latency_sweep", "4x16", "encode_ms"),
        "nq8_encode_ms": _get(full, "query_path", "nq8", "encode_ms"),
        "nq8_encode_frac": _get(full, "query_path", "nq8", "encode_roofline", "frac"),
        "nq64_encode_ms": _get(full, "query_path", "nq64", "encode_ms"),
        "encode_chunks_per_s": _get(full, "encode", "value"),
        "encode_frac": _get(full, "encode", "roofline", "frac"),
        "backfill_chunks_per_s": _get(full, "encode", "backfill_path", "device_resident", "chunks_per_s"),
    }
    out["summary"] = {k: v for k, v in summ.items() if v is not None}
    errs = [name for name in ("hybrid", "query_path", "encode", "target_1m") if _get(full, name, "error")]
    if errs:
        out["leg_errors"] = errs
    out["detail"] = "DETAIL line above; gpurun_out/bench_detail.json"
    text = json.dumps(out)
    if len(text) >= COMPACT_LIMIT:   # never let a long workload string push the contract keys out of the record
        out["config"]["workload"] = str(out["config"]["workload"])[:160]
        out.pop("summary", None)
    return out


def emit(full: dict) -> None:
    """DETAIL first (every leg, one line, prefixed so that no parser takes it for the bench line; also written to
    gpurun_out/bench_detail.json), then the compact line LAST."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as fh:
            json.dump(full, fh)
    except OSError:
        pass
    print("DETAIL " + json.dumps(full), flush=True)
    print(json.dumps(compact_line(full)), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5,
                    help="the K timed steps are repeated this many times; `value` comes from the FIRST round "
                         "(exactly K steps, the contract), median and min over the rounds are reported beside it")
    ap.add_argument("--rows-per-gpu", type=int, default=0, help="override the corpus rows held by each rank")
    ap.add_argument("--mode", choices=("auto", "strong", "weak"), default="auto",
                    help="N > 1: strong = one 1M-row corpus sharded over the ranks (BASELINE configs[2], default); "
                         "weak = 100 000 rows per rank")
    ap.add_argument("--queries", type=int, default=QUERIES_PER_STEP)
    ap.add_argument("--topk", type=int, default=TOPK)
    ap.add_argument("--api", choices=("pipelined", "async"), default="async",
                    help="N = 1: which form of the C ABI the timed steps call (async: crag_index_search_async, in stream "
                         "order -- the default: kernel durations are undisturbed; pipelined: crag_index_search_pipelined + "
                         "one crag_index_join per fence).  The other form is timed beside it.")
    ap.add_argument("--no-other-api", action="store_true",
                    help="skip the leg that times the other form of the API (profiling runs: its launches of the same "
                         "kernel overlap across streams and would be averaged into the rocprofv3 duration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode", action="store_true", help="skip the chunks-embedded/sec leg")
    ap.add_argument("--no-target-1m", action="store_true", help="skip the 1M x 1024 single-GPU leg (N = 1 only)")
    ap.add_argument("--no-overlap-leg", action="store_true",
                    help="skip the 2- and 3-stream repeats of the headline leg (profiling runs: overlapped launches of "
                         "the same kernel would be averaged into its rocprofv3 duration)")
    ap.add_argument("--no-large-k", action="store_true", help="skip the top-50 / top-100 legs")
    ap.add_argument("--no-hybrid", action="store_true", help="skip the configs[4] hybrid leg (N = 1, with target_1m)")
    ap.add_argument("--no-query-path", action="store_true",
                    help="skip the reference-operating-point leg (encode 1/8/64 queries -> top-50 + top-10 searches)")
    ap.add_argument("--encode-steps", type=int, default=20)
    ap.add_argument("--no-fp32-rows-leg", action="store_true",
                    help="skip the legs that repeat the search on an index without the fp16 mirror")
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
    from cadence_rag_amd.sharded import shard_bounds

    nq, k = args.queries, args.topk
    mode = args.mode if args.mode != "auto" else ("strong" if world > 1 else "single")
    if world == 1:
        rows_total = args.rows_per_gpu or ROWS_CONFIG1
        lo, hi = 0, rows_total
        workload = f"BASELINE configs[1]: brute-force cosine top-{k}, {rows_total}x{DIM} fp32 corpus on one GPU, " \
                   f"{nq} queries/step resident in HBM"
    elif mode == "strong":
        rows_total = (args.rows_per_gpu * world) if args.rows_per_gpu else ROWS_CONFIG2
        lo, hi = shard_bounds(rows_total, world, rank)
        workload = f"BASELINE configs[2]: ONE {rows_total}x{DIM} fp32 corpus in {world} contiguous row shards, " \
                   f"{nq} queries/step, per-shard top-{k} + one RCCL all-gather + merge"
    else:
        per = args.rows_per_gpu or ROWS_CONFIG1
        rows_total = per * world
        lo, hi = rank * per, (rank + 1) * per
        workload = f"weak mode: {per}x{DIM} rows per rank ({rows_total} in all), {nq} queries/step, per-shard " \
                   f"top-{k} + one RCCL all-gather + merge"
    rows = hi - lo
    corpus = synth_rows(lo, hi, 1234, dev)  # this rank's rows of the one global corpus
    queries = synth(max(nq, 64), 4321, dev)[:nq].contiguous()  # same queries on every rank
    ids = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    index = DenseIndex(DIM, capacity=rows, device=dev_index)
    index.add(corpus, ids)

    step_extra = fence_extra = outs = None
    if world > 1:
        # the search writes straight into THIS RANK'S SLOT of the gather buffer: the all-gather runs in place
        rec_bytes = ResultRecord.record_bytes(nq, k)
        gathered = torch.zeros(world * rec_bytes, dtype=torch.uint8, device=dev)
        rec = ResultRecord(nq, k, dev, buf=gathered[rank * rec_bytes:(rank + 1) * rec_bytes])
        outs = (rec.ids, rec.scores, rec.counts)
        f_ids = torch.empty(nq, k, dtype=torch.int64, device=dev)
        f_sc = torch.empty(nq, k, dtype=torch.float32, device=dev)
        f_ct = torch.empty(nq, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def step_extra():  # the path's one exchange step: ONE all-gather of 12*Q*k + 4*Q bytes per rank over xGMI,
            # then the merge; everything is enqueued, nothing synchronises with the host
            dist.all_gather_into_tensor(gathered, rec.buf)
            merge_topk_packed(gathered, world, nq, k, f_ids, f_sc, f_ct, stream=stream)

        def fence_extra():
            dist.barrier()

    # N = 1: the timed steps go through crag_index_search_async in stream order (--api async, the default: kernel
    # durations undisturbed for the roofline); the throughput form for a run of independent searches
    # (crag_index_search_pipelined: consecutive searches rotate over three streams of the index's own, one join in
    # front of every fence) is timed beside it.  N > 1: a step's collective consumes the search's output -> in order.
    pipelined = world == 1 and args.api == "pipelined"
    leg = search_leg(index, queries, k, args.steps, args.warmup, args.rounds, step_extra, fence_extra, outs,
                     pipelined=pipelined)
    large_k = {}

    def large_k_leg(ix, q, rows_, kk):
        """The reference's own dense k (retrieve.py:18-19: 50 chunks; configs[4]: 100) on the same corpus and batch."""
        lk = search_leg(ix, q, kk, 200, 10, 2, prewarm_s=0.05)
        return {"ms_per_step": round(min(lk["times"]) / 200 * 1e3, 5), "value": round(int(q.shape[0]) * 200 / min(lk["times"]), 1),
                "unit": "queries/sec", "roofline": roofline(rows_, int(q.shape[0]), kk, lk, None)}

    # (in front of the legs that search this index from several streams: once a second stream has searched an index,
    # every search on it records its workspace's completion event -- 4-7 us per step that a one-stream caller never pays)
    if world == 1 and not args.no_large_k:
        for kk in (50, 100):
            large_k[f"{rows}x{nq}x{kk}"] = large_k_leg(index, queries, rows, kk)
    leg_other = None
    if world == 1 and not args.no_other_api:   # the other form beside it; same bits from both
        leg_other = search_leg(index, queries, k, args.steps, min(args.warmup, 20), min(args.rounds, 3),
                               pipelined=not pipelined, prewarm_s=0.05)
        leg["identical_to_other_api"] = all(bool(torch.equal(a, b)) for a, b in zip(leg["out"], leg_other["out"]))
    times = leg["times"]
    per_rank = None
    if world > 1:
        # where a rank's step goes: local search, the one exchange (all-gather), the merge — torch events on the
        # launch stream around each part of 40 untimed steps (the all-gather's completion is ordered into the
        # stream by ProcessGroupNCCL before the next event is recorded)
        marks = []
        for _ in range(40):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            index.search_async(queries, k, rec.ids, rec.scores, rec.counts, stream=stream)
            ev[1].record()
            dist.all_gather_into_tensor(gathered, rec.buf)
            ev[2].record()
            merge_topk_packed(gathered, world, nq, k, f_ids, f_sc, f_ct, stream=stream)
            ev[3].record()
            marks.append(ev)
        torch.cuda.synchronize()
        mine = {"rank": rank, "rows": rows,
                "search_us": round(statistics.median(e[0].elapsed_time(e[1]) for e in marks) * 1e3, 1),
                "exchange_us": round(statistics.median(e[1].elapsed_time(e[2]) for e in marks) * 1e3, 1),
                "merge_us": round(statistics.median(e[2].elapsed_time(e[3]) for e in marks) * 1e3, 1),
                "scan_kernel_us": round(leg["scan_us"], 1)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        # the merged answer against the fp64 oracle (the checker, outside every timed region): every rank scores a sample of
        # the queries against ITS shard on its host cores, rank 0 merges the per-shard oracle lists (score descending,
        # id ascending -- the search's order) and compares them with what the exchange + merge left on the GPU
        sharded_recall = None
        if not args.no_cpu_baseline:
            import oracle
            oracle.set_threads(max(1, min(os.cpu_count() or 1, 64) // world))
            sample = sorted({0, nq // 3, (2 * nq) // 3, nq - 1})
            t_ids, t_sc, t_ct = oracle.exact_topk(queries.cpu().numpy()[sample], corpus.cpu().numpy(), k, mode=oracle.F64, fast=True)
            mine_truth = [(t_sc[i, :t_ct[i]].tolist(), (t_ids[i, :t_ct[i]] + lo).tolist()) for i in range(len(sample))]
            all_truth = [None] * world
            dist.all_gather_object(all_truth, mine_truth)
            if rank == 0:
                got = f_ids.cpu().numpy()
                hits = same = 0
                for i, q in enumerate(sample):
                    pairs = sorted(((-sc, rid) for r in all_truth for sc, rid in zip(*r[i])))[:k]
                    want = [rid for _, rid in pairs]
                    hits += len(set(want[:10]) & set(got[q, :10].tolist()))
                    same += int(want == got[q, :len(want)].tolist())
                sharded_recall = {"recall_at_10_vs_fp64_oracle": hits / float(len(sample) * min(k, 10)),
                                  "topk_order_identical_to_oracle": same == len(sample), "queries_checked": len(sample)}
    if world > 1:  # max over ranks, per round
        t = torch.tensor(times, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = t.tolist()
    elapsed = times[0]

    traffic_doc = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic_doc = json.load(open(tpath))
        except Exception:
            traffic_doc = None

    overlap = fp32_leg = None
    if world == 1:
        if not args.no_overlap_leg:
            overlap = {str(n): overlap_leg(index, queries, k, n, args.steps) for n in (2, 3)}
        if index.prefilter_row_bytes() == DIM * 2 and not args.no_fp32_rows_leg:
            fp32_leg = fp32_rows_leg(corpus, ids, queries, k, dev_index, max(200, args.steps // 4), leg["out"], traffic_doc)
    target = hybrid = query_path = shared_enc = None
    if world == 1 and not args.no_target_1m and rows_total == ROWS_CONFIG1:
        gpu_ids_100k = leg["out"][0].cpu().numpy()
        corpus_host_100k = corpus.cpu().numpy()
        index.close()
        del corpus, index
        torch.cuda.empty_cache()
        big = synth(ROWS_CONFIG2, 1234, dev)
        big_index = DenseIndex(DIM, capacity=ROWS_CONFIG2, device=dev_index)
        big_index.add(big)
        target = {"workload": f"north-star target: {ROWS_CONFIG2}x{DIM} fp32 corpus on ONE GPU, exact top-{k}, "
                              "queries resident in HBM"}
        q64 = synth(64, 4321, dev)
        big_host = None
        for name, qn, st_ in (("q32", 32, 300), ("q64", 64, 300)):
            tl = search_leg(big_index, q64[:qn].contiguous(), k, st_, 20, 3)
            if big_host is None and not args.no_cpu_baseline:
                big_host = big.cpu().numpy()
            entry = {"queries_per_step": qn, "steps": st_, "rounds": 3,
                     "ms_per_step": round(tl["times"][0] / st_ * 1e3, 5),
                     "ms_per_step_median": round(statistics.median(tl["times"]) / st_ * 1e3, 5),
                     "value": round(qn * st_ / tl["times"][0], 2), "unit": "queries/sec",
                     "roofline": roofline(ROWS_CONFIG2, qn, k, tl, traffic_doc)}
            if big_host is not None:
                sample = [0, qn // 2, qn - 1]
                rc, same = recall_check(big_host, q64.cpu().numpy(), tl["out"][0].cpu().numpy(), k, sample)
                entry["recall_at_10_vs_fp64_oracle"] = rc
                entry["topk_order_identical_to_oracle"] = same
                entry["oracle_sample"] = f"queries {sample} of the step's batch, full {ROWS_CONFIG2}-row fp64 scan each"
            if big_index.prefilter_row_bytes() == DIM * 2 and not args.no_fp32_rows_leg:
                entry["fp32_rows_scan"] = fp32_rows_leg(big, None, q64[:qn].contiguous(), k, dev_index, 150, tl["out"],
                                                        traffic_doc)
            target[name] = entry
        if not args.no_large_k:
            for kk in (50, 100):
                large_k[f"{ROWS_CONFIG2}x64x{kk}"] = large_k_leg(big_index, q64, ROWS_CONFIG2, kk)
        if not args.no_hybrid:
            try:
                hybrid = hybrid_leg(big_index, ROWS_CONFIG2, dev)
            except Exception as exc:  # the headline line must still be printed
                hybrid = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_encode and not args.no_query_path:
            from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
            shared_enc = Qwen3Encoder.random_init(Qwen3Config(), seed=1234, device=dev)
            try:
                query_path = query_path_leg(shared_enc, shared_enc.cfg, big_index, big, dev, dev_index)
            except Exception as exc:
                query_path = {"error": f"{type(exc).__name__}: {exc}"}
        big_index.close()
        del big, big_index
        torch.cuda.empty_cache()
    else:
        gpu_ids_100k = leg["out"][0].cpu().numpy() if world == 1 else None
        corpus_host_100k = corpus.cpu().numpy() if (world == 1 and not args.no_cpu_baseline) else None
        index.close()

    encode = None
    if not args.no_encode:
        encode = encode_leg(dev, rank, world, dist, args.encode_steps, enc=shared_enc)

    same_job_1gpu = None
    if world > 1 and rank == 0 and mode == "strong":
        # the 1-GPU point of THIS job, measured in this run on rank 0's GPU while the other ranks wait at the final
        # barrier: the driver's N = 1 line times configs[1] (100 000 rows), not the 1M-row job that is sharded here
        try:
            whole = synth_rows(0, rows_total, 1234, dev)
            whole_index = DenseIndex(DIM, capacity=rows_total, device=dev_index)
            whole_index.add(whole)
            st1 = max(20, min(args.steps, 300))
            one = search_leg(whole_index, queries, k, st1, min(args.warmup, 20), 1)
            same_job_1gpu = {"rows": rows_total, "steps": st1, "ms_per_step": round(one["times"][0] / st1 * 1e3, 5),
                             "value": round(nq * st1 / one["times"][0], 2), "unit": "queries/sec",
                             "scan_kernel_us": round(one["scan_us"], 1)}
            whole_index.close()
            del whole, whole_index
        except Exception as exc:  # the line must still be printed
            same_job_1gpu = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        roof = roofline(rows, nq, k, leg, traffic_doc)
        line = {
            "metric": "queries/sec @ recall@10=1.0 (exact cosine top-10, 1024-d fp32 corpus)",
            "value": round(nq * args.steps / elapsed, 2),
            "unit": "queries/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong" if mode == "strong" else "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "rows_per_gpu": rows, "rows_total": rows_total, "dim": DIM, "k": k, "queries_per_step": nq,
                "parallelism": "1 GPU" if world == 1 else f"corpus sharded x{world}, all-gather top-k merge",
                # `value` = distinct queries answered per second by the whole job (never multiplied by the rank count)
                "rounds": args.rounds,
                "ms_per_step_rounds": [round(t / args.steps * 1e3, 5) for t in times],
                "ms_per_step_median": round(statistics.median(times) / args.steps * 1e3, 5),
                "ms_per_step_min": round(min(times) / args.steps * 1e3, 5),
                "row_queries_per_s": round(rows_total * nq * args.steps / elapsed, 1),
                # fp16 MFMA prefilter over the fp16 mirror with a proven bound + exact fp32 rescoring from the fp32 rows:
                # results bit-identical to the fp32 scan
                "arithmetic": "fp16-MFMA prefilter (proven bound) + exact fp32 rescoring",
                "api": ("crag_index_search_pipelined + crag_index_join per fence" if pipelined else
                        "crag_index_search_async (in stream order)"),
                "hbm_bytes_per_corpus_row": DIM * 4 + (leg.get("row_bytes") if leg.get("row_bytes") != DIM * 4 else 0) + 12,
            },
            "roofline": roof,
        }
        if world > 1:
            line["config"]["per_rank_step_breakdown"] = per_rank
            if sharded_recall is not None:   # (the sharded job's answer against the per-shard fp64 oracle lists, merged)
                line["config"]["recall_at_10_vs_fp64_oracle"] = sharded_recall["recall_at_10_vs_fp64_oracle"]
                line["config"]["topk_order_identical_to_oracle"] = sharded_recall["topk_order_identical_to_oracle"]
                line["config"]["oracle_queries_checked"] = sharded_recall["queries_checked"]
            if same_job_1gpu is not None:
                line["config"]["same_job_on_one_gpu"] = same_job_1gpu
                if "value" in same_job_1gpu:
                    line["config"]["speedup_vs_same_job_on_one_gpu"] = round(
                        nq * args.steps / elapsed / same_job_1gpu["value"], 3)
            # strong: the 1-GPU point of the fixed 1M-row job is config.same_job_on_one_gpu (= target_1m.q64 of the N = 1
            # line; the N = 1 `value` itself is configs[1], 100 000 rows); weak: a flat `value` is ideal
            line["config"]["scaling_curve_measured"] = "this line is one point; no curve without a multi-GPU node"
        if leg_other is not None:
            t_o = leg_other["times"]
            name = "in_order_api" if pipelined else "pipelined_api"
            line["config"]["identical_to_other_api"] = leg.get("identical_to_other_api")
            line[name] = {"api": ("crag_index_search_async" if pipelined else
                                  "crag_index_search_pipelined + crag_index_join per fence (three streams of the index's own)"),
                          "value": round(nq * args.steps / t_o[0], 2), "unit": "queries/sec",
                          "ms_per_step": round(t_o[0] / args.steps * 1e3, 5),
                          "ms_per_step_median": round(statistics.median(t_o) / args.steps * 1e3, 5),
                          "outputs_identical_on_both_streams": (leg_other if not pipelined else leg).get(
                              "outputs_identical_on_both_streams"),
                          "roofline": roofline(rows, nq, k, leg_other, traffic_doc)}
        if overlap is not None:
            line["config"]["steps_overlapped_on_streams"] = overlap
        if fp32_leg is not None:
            line["fp32_rows_scan"] = fp32_leg
        if large_k:
            line["large_k"] = large_k
        if target is not None:
            line["target_1m"] = target
        if hybrid is not None:
            line["hybrid"] = hybrid
        if query_path is not None:
            line["query_path"] = query_path
        if encode is not None:
            line["encode"] = encode
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (rank 0's host cores)
            queries_host = queries.cpu().numpy()
            base = cpu_baseline(corpus_host_100k, queries_host)
            sample = list(range(0, nq, max(1, nq // 16)))
            recall, same_order = recall_check(corpus_host_100k, queries_host, gpu_ids_100k, k, sample)
            if encode is not None:
                base["encode"] = cpu_encode_baseline()
            line["cpu_baseline"] = base
            line["config"]["recall_at_10_vs_fp64_oracle"] = recall
            line["config"]["topk_order_identical_to_oracle"] = same_order
        emit(line)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
