"""Developer probe: the q/k head-norm + RoPE pass at the bench's shape (65 588 tokens, 32 q + 8 kv heads of 128)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cadence_rag_amd.encoder import ops
from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
dev = torch.device("cuda", 0)
c = Qwen3Config()
T = 65588
qkv = torch.randn(T + 32, (c.num_heads + 2 * c.num_kv_heads) * c.head_dim, device=dev, dtype=torch.bfloat16)
qw = torch.ones(c.head_dim, device=dev, dtype=torch.bfloat16); kw = torch.ones(c.head_dim, device=dev, dtype=torch.bfloat16)
cos_sin = Qwen3Encoder._rope_table(c).to(dev)
pos = (torch.arange(T, device=dev, dtype=torch.int32) % 256).contiguous()
for _ in range(5): ops.qk_norm_rope(qkv, qw, kw, cos_sin, pos, c.num_heads, c.num_kv_heads, c.rms_norm_eps)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): ops.qk_norm_rope(qkv, qw, kw, cos_sin, pos, c.num_heads, c.num_kv_heads, c.rms_norm_eps)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
print(f"qk_norm_rope {dt*1e3:.3f} ms = {T*40*128*2*2/dt/1e12:.2f} TB/s")
