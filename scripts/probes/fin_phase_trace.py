"""Developer probe: where a selection block of finalize_fb_kernel spends its time (CRAG_PHASE_TRACE=1: device timestamps
of query 0's selection blocks at the phase boundaries, 100 MHz)."""
import os, sys, torch
os.environ["CRAG_PHASE_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cadence_rag_amd.dense_index import DenseIndex
dev = torch.device("cuda", 0)
names = ["loaded", "kth", "rescored", "own list+ticket", "gathered", "end"]
for rows in (100_000, 1_000_000):
    big = bench.synth(rows, 1234, dev); idx = DenseIndex(bench.DIM, capacity=rows, device=0); idx.add(big)
    q = bench.synth(64, 4321, dev)
    for k in (10, 50, 100):
        oi = torch.empty(64, k, dtype=torch.int64, device=dev); osc = torch.empty(64, k, dtype=torch.float32, device=dev)
        oc = torch.empty(64, dtype=torch.int32, device=dev)
        for _ in range(30):
            idx.search_async(q, k, oi, osc, oc, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        t = idx.phase_trace()
        print(f"rows {rows} k {k}:")
        t0 = min(t[r * 16] for r in range(8) if t[r * 16])
        for r in range(8):
            row = t[r * 16:r * 16 + 16]
            if not row[0]:
                continue
            marks = [row[0]] + [x for x in row[1:7]]
            txt = [f"start +{(row[0] - t0) / 100:.2f}"]
            prev = row[0]
            for i, nm in enumerate(names):
                x = row[i + 1]
                if x and x >= prev:       # (a mark older than the block's start is left over from an earlier search:
                                          # only the block that arrives last gathers)
                    txt.append(f"{nm} {(x - prev) / 100:.2f}")
                    prev = x
            print(f"  block {r}: " + ", ".join(txt) + f"  | total {(prev - row[0]) / 100:.2f} us, C={row[8]}, rescored={row[9]}")
    idx.close(); del big
