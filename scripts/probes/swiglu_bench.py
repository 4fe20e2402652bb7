"""Developer probe: the SwiGLU pass at the bench's shape (65 588 tokens x 9 728)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops
dev = torch.device("cuda", 0)
T, I = 65588, 9728
gu = torch.randn(T, 2 * I, device=dev, dtype=torch.bfloat16)
out = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
for _ in range(5): ops.swiglu(gu, out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): ops.swiglu(gu, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
ref = (torch.nn.functional.silu(gu[:64, :I].float()).to(torch.bfloat16).float() * gu[:64, I:].float()).to(torch.bfloat16)
print(f"swiglu {dt*1e3:.3f} ms = {T*I*2*3/dt/1e12:.2f} TB/s, max |d| vs torch on 64 rows {float((out[:64].float()-ref.float()).abs().max()):.3e}")
