"""Developer probe: time DenseIndex.search_async for a few (rows, queries, k) shapes and print per-kernel event
times and the prefilter path's candidate statistics.   python scripts/probes/search_bench.py 100000,64,10 ..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.dense_index import DenseIndex  # noqa: E402


def run(rows: int, nq: int, k: int, steps: int = 200) -> None:
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    index = DenseIndex(1024, capacity=rows)
    for lo in range(0, rows, 131072):
        m = min(131072, rows - lo)
        c = torch.randn(m, 1024, generator=g, device=dev)
        index.add(c / c.norm(dim=1, keepdim=True))
    q = torch.randn(nq, 1024, generator=g, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    oi = torch.empty(nq, k, dtype=torch.int64, device=dev)
    osc = torch.empty(nq, k, dtype=torch.float32, device=dev)
    oc = torch.empty(nq, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        index.search_async(q, k, oi, osc, oc, stream=st)
    torch.cuda.synchronize()
    # step time without any profiling events in the stream, at 20 steps per sync (the driver's setting) and at `steps`
    dts = {}
    for n in (20, steps):
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                index.search_async(q, k, oi, osc, oc, stream=st)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        dts[n] = best
    index.prefilter_stats()
    index.profile_enable(4)
    t0 = time.perf_counter()
    for _ in range(steps):
        index.search_async(q, k, oi, osc, oc, stream=st)
    enq = (time.perf_counter() - t0) / steps      # host time to enqueue one search (the GPU may lag behind)
    torch.cuda.synchronize()
    dt = dts[steps]
    n, scan_ms, rest_ms = index.profile_read()
    index.profile_enable(0)
    stats = index.prefilter_stats()
    per = max(stats["searches"], 1)
    row_bytes = index.prefilter_row_bytes() if "prefilter" in index.last_scan_kernel() else 4096
    bytes_ = rows * row_bytes + nq * 4096
    scan_us = scan_ms / max(n, 1) * 1e3
    print(f"rows={rows} nq={nq} k={k}: step {dt * 1e6:8.1f} us ({dts[20] * 1e6:6.1f} at 20 steps/sync; host enqueue {enq * 1e6:5.1f})  {nq / dt:10.0f} q/s | {index.last_scan_kernel()} "
          f"{scan_us:8.1f} us = {bytes_ / scan_us / 1e3:6.0f} GB/s ({bytes_ / scan_us / 1e3 / 8000:.3f} of 8 TB/s), "
          f"rest {rest_ms / max(n, 1) * 1e3:6.1f} us | cand/search {stats['candidates'] / per:8.1f} "
          f"rescored/search {stats['rescored_rows'] / per:8.1f}", flush=True)
    index.close()


if __name__ == "__main__":
    shapes = sys.argv[1:] or ["100000,64,10", "100000,32,10", "1000000,32,10", "1000000,64,10", "1000000,64,100"]
    for s in shapes:
        r, nq, k = (int(v) for v in s.split(","))
        run(r, nq, k)
