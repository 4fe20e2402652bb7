"""Developer probe: the RMSNorm(+residual) pass at the bench's shape (65 588 tokens x 2 560)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops
dev = torch.device("cuda", 0)
T, H = 65588, 2560
x = torch.randn(T, H, device=dev, dtype=torch.bfloat16); res = torch.randn(T, H, device=dev, dtype=torch.bfloat16)
w = torch.ones(H, device=dev, dtype=torch.bfloat16); out = torch.empty_like(x)
for _ in range(5): ops.rmsnorm(x, w, out, 1e-6, residual_in=res, residual_out=res)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): ops.rmsnorm(x, w, out, 1e-6, residual_in=res, residual_out=res)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
print(f"rmsnorm+residual {dt*1e3:.3f} ms = {T*H*2*4/dt/1e12:.2f} TB/s")
