"""Encoder lane: every hand-written HIP operator against a plain PyTorch fp32 reference of the same
op, and the whole packed forward against transformers' Qwen3Model (fp32, CPU) built from a local
config with seeded random weights — the only encoder oracle available offline."""
import math

import numpy as np
import pytest
import torch

from cadence_rag_amd.dense_index import DenseIndex

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
BF = torch.bfloat16


def _bf(x):
    return x.to(BF).to(DEV).contiguous()


def test_embed_gather(gpu):
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(0)
    table = _bf(torch.randn(1000, 256, generator=g))
    ids = torch.randint(0, 1000, (77,), generator=g, dtype=torch.int32).to(DEV)
    out = torch.empty(77, 256, dtype=BF, device=DEV)
    ops.embed_gather(ids, table, out)
    assert torch.equal(out, table[ids.long()])


@pytest.mark.parametrize("rows,hidden", [(1, 8), (5, 256), (33, 2560), (7, 8192)])
def test_rmsnorm_with_and_without_residual(gpu, rows, hidden):
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(rows * 31 + hidden)
    x, r = _bf(torch.randn(rows, hidden, generator=g) * 3), _bf(torch.randn(rows, hidden, generator=g))
    w = _bf(1 + 0.1 * torch.randn(hidden, generator=g))
    eps = 1e-6

    def ref(s_bf):
        s = s_bf.float()
        n = (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + eps)).to(BF)
        return (w.float() * n.float()).to(BF)

    out = torch.empty_like(x)
    ops.rmsnorm(x, w, out, eps)
    assert torch.allclose(out.float(), ref(x).float(), atol=2e-2, rtol=2e-2)
    res_out = torch.empty_like(x)
    ops.rmsnorm(x, w, out, eps, residual_in=r, residual_out=res_out)
    s = (x.float() + r.float()).to(BF)
    assert torch.equal(res_out, s)
    assert torch.allclose(out.float(), ref(s).float(), atol=2e-2, rtol=2e-2)


def _rope_ref(x, pos, theta):
    # x [T, H, 128] fp32, rotate-half form
    half = 64
    inv = 1.0 / (theta ** (torch.arange(0, half, dtype=torch.float32) * 2.0 / 128))
    ang = pos.float()[:, None] * inv[None, :]
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1)[:, None, :], torch.cat([ang.sin(), ang.sin()], -1)[:, None, :]
    rot = torch.cat([-x[..., half:], x[..., :half]], -1)
    return x * cos + rot * sin


def test_qk_norm_rope(gpu):
    from cadence_rag_amd.encoder import ops
    from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
    g = torch.Generator().manual_seed(3)
    t, hq, hkv = 37, 4, 2
    qkv = _bf(torch.randn(t + 32, (hq + 2 * hkv) * 128, generator=g))
    qw, kw = _bf(1 + 0.1 * torch.randn(128, generator=g)), _bf(1 + 0.1 * torch.randn(128, generator=g))
    pos = torch.randint(0, 500, (t,), generator=g, dtype=torch.int32)
    cfg = Qwen3Config(max_length=512)
    table = Qwen3Encoder._rope_table(cfg).to(DEV)
    before = qkv.clone()
    ops.qk_norm_rope(qkv, qw, kw, table, pos.to(DEV), hq, hkv, 1e-6)

    def ref(block, w, heads):
        v = block.float().cpu().view(t, heads, 128)
        n = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float().cpu()
        return _rope_ref(n, pos, cfg.rope_theta).reshape(t, heads * 128)

    q_ref = ref(before[:t, : hq * 128], qw, hq)
    k_ref = ref(before[:t, hq * 128: (hq + hkv) * 128], kw, hkv)
    assert torch.allclose(qkv[:t, : hq * 128].float().cpu(), q_ref, atol=3e-2, rtol=3e-2)
    assert torch.allclose(qkv[:t, hq * 128: (hq + hkv) * 128].float().cpu(), k_ref, atol=3e-2, rtol=3e-2)
    assert torch.equal(qkv[:, (hq + hkv) * 128:], before[:, (hq + hkv) * 128:])  # V untouched
    assert torch.equal(qkv[t:], before[t:])  # rows past T untouched


def _attn_ref(q, k, v, lens, hq, hkv):
    # q [T, hq, 128], k/v [T, hkv, 128] fp32; causal per sequence, GQA
    out = torch.zeros_like(q)
    start = 0
    for n in lens:
        for h in range(hq):
            kv = h // (hq // hkv)
            s = (q[start:start + n, h] @ k[start:start + n, kv].T) / math.sqrt(128)
            s = s.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool), 1), float("-inf"))
            out[start:start + n, h] = torch.softmax(s, -1) @ v[start:start + n, kv]
        start += n
    return out


@pytest.mark.parametrize("lens,hq,hkv", [([1], 4, 1), ([31, 32, 33], 4, 2), ([5, 200, 64, 1, 97], 8, 2),
                                         ([300, 17], 32, 8), ([40, 3, 130], 2, 2), ([70, 33], 8, 1)])
def test_v_transpose_and_attention(gpu, lens, hq, hkv):
    from cadence_rag_amd.encoder import ops
    from cadence_rag_amd.encoder.qwen3 import PackedBatch
    g = torch.Generator().manual_seed(sum(lens) + hq)
    t = sum(lens)
    width = (hq + 2 * hkv) * 128
    qkv = torch.randn(t + 32, width, generator=g)
    qkv[t:] = float("nan")  # rows past T may hold anything
    qkv = _bf(qkv)
    batch = PackedBatch.build(lens, DEV)
    vt = torch.empty(hkv, 128, batch.t_pad, dtype=BF, device=DEV)
    ops.v_transpose(qkv, vt, batch.tok_of_pad, hq, hkv)
    v = qkv[:t, (hq + hkv) * 128:].view(t, hkv, 128)
    tok = batch.tok_of_pad.long()
    want_vt = torch.zeros(hkv, 128, batch.t_pad, dtype=BF, device=DEV)
    want_vt[:, :, tok >= 0] = v[tok[tok >= 0]].permute(1, 2, 0)
    # inside each 32-slot block: stored index 16*s2 + 8*h + 4*g + r holds slot 16*s2 + 8*g + 4*h + r
    st = torch.arange(32)
    src = 16 * (st >> 4) + 8 * ((st >> 2) & 1) + 4 * ((st >> 3) & 1) + (st & 3)
    perm = (torch.arange(batch.t_pad).view(-1, 32)[:, :1] + src.view(1, 32)).reshape(-1).to(DEV)
    assert torch.equal(vt, want_vt[:, :, perm])
    out = torch.empty(t, hq * 128, dtype=BF, device=DEV)
    ops.attention(qkv, vt, out, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0, hq, hkv, 1 / math.sqrt(128))
    f = qkv[:t].float().cpu()
    ref = _attn_ref(f[:, : hq * 128].view(t, hq, 128), f[:, hq * 128: (hq + hkv) * 128].view(t, hkv, 128),
                    f[:, (hq + hkv) * 128:].view(t, hkv, 128), lens, hq, hkv)
    got = out.float().cpu().view(t, hq, 128)
    assert torch.isfinite(got).all()
    assert torch.allclose(got, ref, atol=2e-2, rtol=2e-2), (got - ref).abs().max()


def test_attention_rescale_branch_with_spiked_key(gpu):
    """Force a large running-max jump at a late key tile (cdna guide rule 26)."""
    from cadence_rag_amd.encoder import ops
    from cadence_rag_amd.encoder.qwen3 import PackedBatch
    g = torch.Generator().manual_seed(9)
    lens, hq, hkv = [160], 4, 1
    t = 160
    qkv = torch.randn(t + 32, (hq + 2 * hkv) * 128, generator=g) * 0.5
    qkv[130, hq * 128: (hq + 1) * 128] = qkv[150, :128] * 6.0  # key 130 aligned with query 150 of head 0
    qkv = _bf(qkv)
    batch = PackedBatch.build(lens, DEV)
    vt = torch.empty(hkv, 128, batch.t_pad, dtype=BF, device=DEV)
    ops.v_transpose(qkv, vt, batch.tok_of_pad, hq, hkv)
    out = torch.empty(t, hq * 128, dtype=BF, device=DEV)
    ops.attention(qkv, vt, out, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0, hq, hkv, 1 / math.sqrt(128))
    f = qkv[:t].float().cpu()
    ref = _attn_ref(f[:, : hq * 128].view(t, hq, 128), f[:, hq * 128: (hq + hkv) * 128].view(t, hkv, 128),
                    f[:, (hq + hkv) * 128:].view(t, hkv, 128), lens, hq, hkv)
    assert torch.allclose(out.float().cpu().view(t, hq, 128), ref, atol=2e-2, rtol=2e-2)


def test_swiglu(gpu):
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(4)
    gu = _bf(torch.randn(19, 2 * 9728, generator=g) * 2)
    out = torch.empty(19, 9728, dtype=BF, device=DEV)
    ops.swiglu(gu, out)
    gate, up = gu[:, :9728].float(), gu[:, 9728:].float()
    ref = (torch.nn.functional.silu(gate).to(BF).float() * up).to(BF)
    assert torch.allclose(out.float(), ref.float(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("mode", [0, 1])
def test_pool_normalize(gpu, mode):
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(5)
    lens = [3, 1, 10]
    hs = _bf(torch.randn(sum(lens), 256, generator=g))
    w = _bf(1 + 0.1 * torch.randn(256, generator=g))
    cu = torch.tensor([0, 3, 4, 14], dtype=torch.int32, device=DEV)
    out = torch.empty(3, 64, dtype=torch.float32, device=DEV)
    ops.pool_normalize(hs, w if mode == 0 else None, cu, out, 64, mode, 1e-6)
    rows = []
    for b in range(3):
        seg = hs[cu[b]:cu[b + 1]].float()
        if mode == 0:
            x = seg[-1]
            x = (w.float() * (x * torch.rsqrt(x.pow(2).mean() + 1e-6)).to(BF).float()).to(BF).float()
        else:
            x = seg.mean(0)
        x = x[:64]
        rows.append(x / x.norm().clamp_min(1e-12))
    ref = torch.stack(rows)
    assert torch.allclose(out, ref, atol=5e-3, rtol=5e-3)
    assert torch.allclose(out.norm(dim=1), torch.ones(3, device=DEV), atol=1e-5)


def test_pool_normalize_adds_the_last_delta_for_the_pooled_rows(gpu):
    """Last-token pooling with the final residual add folded in: the same bits as adding over every token first."""
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(6)
    hs = _bf(torch.randn(14, 256, generator=g))
    delta = _bf(0.3 * torch.randn(14, 256, generator=g))
    w = _bf(1 + 0.1 * torch.randn(256, generator=g))
    cu = torch.tensor([0, 3, 4, 14], dtype=torch.int32, device=DEV)
    fused = torch.empty(3, 64, dtype=torch.float32, device=DEV)
    plain = torch.empty(3, 64, dtype=torch.float32, device=DEV)
    ops.pool_normalize(hs, w, cu, fused, 64, 0, 1e-6, delta=delta)
    ops.pool_normalize(hs + delta, w, cu, plain, 64, 0, 1e-6)   # torch's bf16 add over all tokens
    assert torch.equal(fused, plain)
    with pytest.raises(Exception):
        ops.pool_normalize(hs, None, cu, fused, 64, 1, 1e-6, delta=delta)


def _tiny_hf_and_mine(pooling="last"):
    from transformers import Qwen3Config as HFConfig
    from transformers.models.qwen3.modeling_qwen3 import Qwen3Model
    from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
    torch.manual_seed(1234)
    hf_cfg = HFConfig(vocab_size=503, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                      num_attention_heads=4, num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-6,
                      max_position_embeddings=1024, rope_parameters={"rope_theta": 1_000_000.0, "rope_type": "default"},
                      attention_bias=False, tie_word_embeddings=False)
    model = Qwen3Model(hf_cfg).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "norm" in name:
                p.copy_(1 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) * 0.05)
            p.copy_(p.to(BF).float())  # the encoder holds bf16 weights: give the oracle the same values
    cfg = Qwen3Config(hidden_size=256, num_layers=3, num_heads=4, num_kv_heads=2, head_dim=128, intermediate_size=512,
                      vocab_size=503, out_dim=64, pooling=pooling, max_length=1024)
    enc = Qwen3Encoder.from_state_dict(cfg, model.state_dict(), DEV)
    return model, enc, cfg


def _hf_embed(model, cfg, token_lists, pooling):
    outs = []
    with torch.no_grad():
        for ids in token_lists:  # one unpadded sequence at a time: the gateway's batch-of-1 case
            h = model(input_ids=torch.tensor([ids])).last_hidden_state[0]
            v = (h[-1] if pooling == "last" else h.mean(0))[: cfg.out_dim].float()
            outs.append(v / v.norm().clamp_min(1e-12))
    return torch.stack(outs)


def test_pool_normalize_of_rows_given_as_data_equals_gather_then_pool(gpu):
    """crag_enc_pool_normalize_rows (the pooled rows as an int64 device array: what a graph replay over padded sequences
    needs) against index_select + crag_enc_pool_normalize_add over one-row sequences: the same bits."""
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(77)
    t, hidden, out_dim = 40, 2560, 1024
    hs, delta = _bf(torch.randn(t, hidden, generator=g)), _bf(torch.randn(t, hidden, generator=g) * 0.1)
    w = _bf(1 + 0.1 * torch.randn(hidden, generator=g))
    rows = torch.tensor([39, 0, 17, 17, 5], dtype=torch.int64, device=DEV)
    cu_one = torch.arange(rows.numel() + 1, dtype=torch.int32, device=DEV)
    for d in (delta, None):
        want = torch.empty(rows.numel(), out_dim, dtype=torch.float32, device=DEV)
        ops.pool_normalize(hs.index_select(0, rows), w, cu_one, want, out_dim, 0, 1e-6,
                           delta=None if d is None else d.index_select(0, rows))
        got = torch.full((rows.numel() + 1, out_dim), 3.0, dtype=torch.float32, device=DEV)
        ops.pool_normalize_rows(hs, w, rows, got[:rows.numel()], out_dim, 1e-6, delta=d)
        assert torch.equal(got[:rows.numel()], want) and torch.all(got[rows.numel()] == 3.0)


@pytest.mark.parametrize("pooling", ["last", "mean"])
def test_full_forward_matches_transformers_qwen3(gpu, pooling):
    model, enc, cfg = _tiny_hf_and_mine(pooling)
    rng = np.random.default_rng(7)
    token_lists = [rng.integers(0, 503, size=n).tolist() for n in (1, 7, 32, 33, 150, 64, 257)]
    got = enc.embed_token_lists(token_lists).cpu()
    want = _hf_embed(model, cfg, token_lists, pooling)
    cos = (got * want).sum(-1)
    assert torch.allclose(got.norm(dim=1), torch.ones(len(token_lists)), atol=1e-4)
    # bf16 pipeline vs fp32 oracle, per element of a unit-norm 64-d output (typical magnitude 1/8): ~36 bf16
    # roundings (3 layers x 12) of rms 2^-8/sqrt(3) = 2.3e-3 relative each add to sqrt(36) * 2.3e-3 = 1.4 % =
    # 1.7e-3 rms; measured on MI355X 1.35e-3 rms / 4.3e-3 max (last), 6.8e-4 / 2.4e-3 (mean); bars ~2x measured
    diff = (got - want).abs()
    print(f"\ntoy forward ({pooling}): max |d| = {diff.max():.2e}, rms = {diff.pow(2).mean().sqrt():.2e}, "
          f"min cos = {cos.min():.6f}")
    assert cos.min() > 0.9995, cos
    assert diff.pow(2).mean().sqrt() < 2.5e-3
    assert diff.max() < 1e-2


def test_last_layer_on_pooled_rows_only_equals_the_full_last_layer(gpu, monkeypatch):
    """Last-token pooling: the last layer's output projection and MLP run on the pooled rows only.  Same math;
    the GEMMs of B rows and of T rows may round the last bf16 bit differently (other tile shapes)."""
    model, enc, cfg = _tiny_hf_and_mine("last")
    rng = np.random.default_rng(17)
    token_lists = [rng.integers(0, 503, size=n).tolist() for n in (1, 2, 31, 32, 33, 100, 257, 64)]
    short = enc.embed_token_lists(token_lists)
    monkeypatch.setenv("CRAG_ENC_FULL_LAST_LAYER", "1")
    full = enc.embed_token_lists(token_lists)
    monkeypatch.delenv("CRAG_ENC_FULL_LAST_LAYER")
    diff = (short - full).abs()
    print(f"\nlast layer on pooled rows vs full: max |d| = {diff.max():.2e}")
    assert diff.max() < 2e-3            # a bf16 ulp or two of an element of magnitude 1/8
    assert ((short * full).sum(-1)).min() > 0.99999
    want = _hf_embed(model, cfg, token_lists, "last")
    assert ((short.cpu() * want).sum(-1)).min() > 0.9995


@pytest.mark.parametrize("k,n,swiglu", [(2560, 6144, False), (4096, 2560, False), (9728, 2560, False), (2560, 19456, True)])
@pytest.mark.parametrize("m_rows,m_pad", [(1, 16), (16, 16), (17, 32), (32, 32)])
def test_skinny_gemm_streams_the_weights_for_one_short_query(gpu, k, n, swiglu, m_rows, m_pad):
    """crag_enc_skinny_gemm (the projections of the encoder at 16 / 32 tokens, the reference's one-query-per-request
    operating point) against torch: bf16 operands, fp32 accumulation; the SwiGLU form against crag_enc_swiglu's
    arithmetic on the bf16-rounded reference projection.  Padding rows never reach the output."""
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(k + n + m_rows)
    x = torch.randn(m_pad, k, generator=g).to(BF)
    x[m_rows:] = float("nan")                                   # padding rows: anything, even NaN
    w = (torch.randn(n, k, generator=g) * 0.02).to(BF)
    xd, wd = x.to(DEV), w.to(DEV)
    ref = x[:m_rows].float() @ w.float().t()                    # [m_rows, n] fp32
    if not swiglu:
        out = torch.full((m_rows + 1, n), 7.0, dtype=BF, device=DEV)
        ops.skinny_gemm(xd, ops.skinny_weight(wd), out[:m_rows], m_rows, n)
        got = out[:m_rows].float().cpu()
        assert torch.all(out[m_rows] == 7.0)                    # nothing written behind the real rows
        assert torch.allclose(got, ref, atol=2e-2, rtol=1.2e-2), (got - ref).abs().max()
        assert (got - ref.to(BF).float()).abs().max() <= 2 * ref.abs().max() * 2 ** -8
    else:
        inter = n // 2
        out = torch.full((m_rows + 1, inter), 7.0, dtype=BF, device=DEV)
        ops.skinny_gemm(xd, ops.skinny_gate_up_weight(wd), out[:m_rows], m_rows, n, swiglu=True)
        assert torch.all(out[m_rows] == 7.0)
        want = torch.empty(m_rows, inter, dtype=BF, device=DEV)
        ops.swiglu(ref.to(BF).to(DEV).contiguous(), want)
        got, want = out[:m_rows].float().cpu(), want.float().cpu()
        # gate / up sums may round to a neighbouring bf16 (another summation order): one bf16 ulp through silu * up
        assert torch.allclose(got, want, atol=3e-3, rtol=3e-2), (got - want).abs().max()


def test_rope_and_v_transpose_in_one_launch_equal_the_two_kernels(gpu):
    from cadence_rag_amd.encoder import ops
    from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder
    g = torch.Generator().manual_seed(99)
    hq, hkv = 8, 2
    lens = [5, 40, 1, 33]
    batch = PackedBatch.build(lens, DEV)
    t = batch.n_tokens
    qkv = torch.randn(t + 32, (hq + 2 * hkv) * 128, generator=g).to(BF).to(DEV)
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF).to(DEV)
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(BF).to(DEV)
    cos_sin = Qwen3Encoder._rope_table(Qwen3Config(max_length=64)).to(DEV)
    a, b = qkv.clone(), qkv.clone()
    vt_a = torch.zeros(hkv, 128, batch.t_pad, dtype=BF, device=DEV)
    vt_b = torch.zeros_like(vt_a)
    ops.qk_norm_rope(a, qw, kw, cos_sin, batch.positions, hq, hkv, 1e-6)
    ops.v_transpose(a, vt_a, batch.tok_of_pad, hq, hkv)
    ops.qk_rope_vt(b, qw, kw, cos_sin, batch.positions, hq, hkv, 1e-6, vt_b, batch.tok_of_pad)
    assert torch.equal(a, b) and torch.equal(vt_a, vt_b)


def test_small_batches_replay_one_graph_per_shape_and_match_the_eager_forward(gpu, monkeypatch):
    """A /retrieve request embeds one short query: embed_token_lists pads every sequence behind its last real token
    to a bucket length and replays a captured graph per (sequences, bucket).  Same embeddings as the eager packed
    forward (up to the GEMM tiling the library picks for another M) and as transformers; the pads behind the pooled
    token change nothing; a graph is reused for other lengths of its bucket."""
    model, enc, cfg = _tiny_hf_and_mine("last")
    rng = np.random.default_rng(23)
    cases = [[5], [32], [33], [1], [17, 3, 32, 9], [64, 60, 40], [210], [16] * 8]
    for lens in cases:
        token_lists = [rng.integers(0, 503, size=n).tolist() for n in lens]
        monkeypatch.delenv("CRAG_ENC_NO_GRAPH", raising=False)
        fast = enc.embed_token_lists(token_lists)
        again = enc.embed_token_lists(token_lists)
        assert torch.equal(fast, again)                         # a replay is deterministic
        monkeypatch.setenv("CRAG_ENC_NO_GRAPH", "1")
        eager = enc.embed_token_lists(token_lists)
        monkeypatch.delenv("CRAG_ENC_NO_GRAPH")
        want = _hf_embed(model, cfg, token_lists, "last")
        assert ((fast * eager).sum(-1)).min() > 0.99999 and (fast - eager).abs().max() < 2e-3, lens
        assert ((fast.cpu() * want).sum(-1)).min() > 0.9995, lens
    keys = sorted(enc._graphs)
    assert keys == [(1, 16), (1, 32), (1, 64), (1, 256), (3, 64), (4, 32), (8, 16)], keys
    # beyond the token budget of the graph path the eager packed forward runs (no new graph)
    big = [rng.integers(0, 503, size=600).tolist() for _ in range(5)]
    enc.embed_token_lists(big)
    assert sorted(enc._graphs) == keys


def test_packed_batch_equals_one_by_one_and_encoder_protocol(gpu, monkeypatch):
    from cadence_rag_amd import embeddings
    from cadence_rag_amd.config import settings
    from cadence_rag_amd.encoder.qwen3 import ByteTokenizer
    _, enc, cfg = _tiny_hf_and_mine()
    enc.tokenizer = ByteTokenizer(eos_id=300)
    texts = ["hello world", "a", "the quick brown fox jumps over the lazy dog " * 3]
    batch_vecs, model_id = enc.encode(texts)
    singles = [enc.encode([t])[0][0] for t in texts]
    # packing changes nothing beyond bf16 GEMM rounding (the library picks other tilings per M)
    assert np.allclose(np.array(batch_vecs), np.array(singles), atol=1e-2)
    assert (np.array(batch_vecs) * np.array(singles)).sum(-1).min() > 0.9995
    monkeypatch.setattr(settings, "embeddings_base_url", "native")
    monkeypatch.setattr(settings, "embeddings_dim", cfg.out_dim)
    embeddings.set_encoder(enc)
    try:
        res = embeddings.embed_texts(["  hello world  ", "", "a"])  # blanks dropped, texts stripped
        assert len(res.vectors) == 2 and res.model == model_id
        assert np.allclose(res.vectors[0], batch_vecs[0], atol=1e-2)  # batch of 2 vs 3: other GEMM tiling
    finally:
        embeddings.set_encoder(None)


def test_attention_property_random_lengths_and_groups(gpu):
    """Random packed batches (ragged lengths around the 32/64-key tile boundaries) and GQA group sizes against
    the fp32 reference: covers full pairs, half pairs and the diagonal tile in either half."""
    from hypothesis import HealthCheck, given, settings, strategies as st
    from cadence_rag_amd.encoder import ops
    from cadence_rag_amd.encoder.qwen3 import PackedBatch

    @settings(max_examples=12, deadline=None, derandomize=True,
              suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
    @given(lens=st.lists(st.sampled_from([1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 128, 129, 200]), min_size=1, max_size=4),
           heads=st.sampled_from([(4, 1), (8, 2), (4, 4), (8, 1), (4, 2)]), seed=st.integers(0, 1000))
    def run(lens, heads, seed):
        hq, hkv = heads
        g = torch.Generator().manual_seed(seed)
        t = sum(lens)
        qkv = torch.randn(t + 32, (hq + 2 * hkv) * 128, generator=g)
        qkv[t:] = float("nan")
        qkv = _bf(qkv)
        batch = PackedBatch.build(lens, DEV)
        vt = torch.empty(hkv, 128, batch.t_pad, dtype=BF, device=DEV)
        ops.v_transpose(qkv, vt, batch.tok_of_pad, hq, hkv)
        out = torch.full((t, hq * 128), float("nan"), dtype=BF, device=DEV)
        ops.attention(qkv, vt, out, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0, hq, hkv, 1 / math.sqrt(128))
        f = qkv[:t].float().cpu()
        ref = _attn_ref(f[:, : hq * 128].view(t, hq, 128), f[:, hq * 128: (hq + hkv) * 128].view(t, hkv, 128),
                        f[:, (hq + hkv) * 128:].view(t, hkv, 128), lens, hq, hkv)
        got = out.float().cpu().view(t, hq, 128)
        assert torch.isfinite(got).all()
        assert torch.allclose(got, ref, atol=2e-2, rtol=2e-2), (lens, heads, (got - ref).abs().max())

    run()


# ------------------------------------------------------------------------------------------------------
# the widths the bench times (Qwen3-Embedding-4B: hidden 2560, ffn 9728, 32 q / 8 kv heads x 128)
# ------------------------------------------------------------------------------------------------------
def _real_width_hf_and_mine(layers=2, vocab=2048):
    from transformers import Qwen3Config as HFConfig
    from transformers.models.qwen3.modeling_qwen3 import Qwen3Model
    from cadence_rag_amd.encoder.qwen3 import Qwen3Config, Qwen3Encoder
    torch.manual_seed(4321)
    hf_cfg = HFConfig(vocab_size=vocab, hidden_size=2560, intermediate_size=9728, num_hidden_layers=layers,
                      num_attention_heads=32, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                      max_position_embeddings=1024, rope_parameters={"rope_theta": 1_000_000.0, "rope_type": "default"},
                      attention_bias=False, tie_word_embeddings=False)
    if layers <= 4:
        model = Qwen3Model(hf_cfg).eval()
    else:  # 36 layers = 14.5 GB of fp32 parameters: skip the default initialiser, fill in place below
        with torch.device("meta"):
            model = Qwen3Model(hf_cfg)
        model = model.to_empty(device="cpu").float().eval()
        for mod in model.modules():  # rotary tables are buffers: rebuild them after to_empty
            if hasattr(mod, "inv_freq") and hasattr(mod, "original_inv_freq"):
                inv = 1.0 / (1_000_000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
                mod.inv_freq = inv
                mod.original_inv_freq = inv
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "norm" in name:
                p.copy_(1 + 0.1 * torch.randn_like(p))
            else:
                p.normal_(0.0, 0.02)
            p.copy_(p.to(BF).float())  # both sides hold the same bf16-representable weights
    cfg = Qwen3Config(num_layers=layers, vocab_size=vocab)  # every other field is the 4B default
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim,
            cfg.out_dim) == (2560, 9728, 32, 8, 128, 1024)
    return model, Qwen3Encoder.from_state_dict(cfg, model.state_dict(), DEV), cfg


def test_real_width_layers_match_transformers_qwen3(gpu):
    """Two decoder layers at the exact 4B widths, ragged lengths incl. 1 / 33 / 257 / 1024, packed HIP forward
    vs transformers' Qwen3Model in fp32 on the CPU.  Tolerance, per element of the unit-norm 1024-d output
    (typical magnitude 1/32 = 3.1e-2): the HIP path keeps activations in bf16 — 8 significant bits, unit
    roundoff 2^-8, rms relative rounding error 2^-8/sqrt(3) = 2.3e-3 — and rounds ~12 times per layer (norm
    out, qkv, rope, P, attention out, o-proj, two residual sums, norm, gate|up, SwiGLU, down).  Independent
    roundings add in quadrature: sqrt(24) * 2.3e-3 = 1.1 % of an element's magnitude = 3.5e-4 rms; over
    8 x 1024 outputs the largest is ~4.5 sigma = 1.6e-3.  Measured on MI355X: rms 4.2e-4, max 1.9e-3, min
    cosine 0.99988 (1 - 1024 * rms^2 / 2 = 0.99991).  Bars = 1.5x the measured values: rms <= 6.5e-4,
    max <= 3e-3, cosine >= 0.9998."""
    model, enc, cfg = _real_width_hf_and_mine()
    rng = np.random.default_rng(11)
    lens = (1, 33, 257, 1024, 8, 64, 300, 31)
    token_lists = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in lens]
    got = enc.embed_token_lists(token_lists).cpu()
    want = _hf_embed(model, cfg, token_lists, "last")
    diff = (got - want).abs()
    cos = (got * want).sum(-1)
    print(f"\nreal-width 2-layer forward vs transformers fp32: max |d| = {diff.max():.2e}, "
          f"rms = {diff.pow(2).mean().sqrt():.2e}, min cos = {cos.min():.6f}")
    assert torch.allclose(got.norm(dim=1), torch.ones(len(lens)), atol=1e-5)
    per_seq = diff.pow(2).mean(dim=1).sqrt()
    print("  per sequence (len: rms): " + ", ".join(f"{n}: {v:.1e}" for n, v in zip(lens, per_seq.tolist())))
    assert diff.pow(2).mean().sqrt() <= 6.5e-4
    assert diff.max() <= 3e-3
    assert cos.min() >= 0.9998


def test_full_depth_36_layers_match_transformers_qwen3(gpu):
    """All 36 layers at the exact 4B widths (the model bench.py times), 2 x 64 tokens + a one-token and a 200-token
    sequence, against transformers' Qwen3Model in fp32 on the CPU: how the bf16 roundings compound over the depth.
    A-priori bar, per element of the unit-norm 1024-d output (typical magnitude 1/32): ~12 bf16 roundings per layer
    of rms relative size 2^-8/sqrt(3) = 2.3e-3, independent => sqrt(36 * 12) * 2.3e-3 = 4.7 % of an element =
    1.5e-3 rms.  The test allows twice that (3e-3 rms; a network that amplifies perturbations from layer to layer
    would exceed it), max <= 5 sigma of the bar = 1.5e-2, cosine >= 1 - 1024 * (3e-3)^2 / 2 = 0.9954."""
    model, enc, cfg = _real_width_hf_and_mine(layers=36)
    assert len(enc.layers) == 36
    rng = np.random.default_rng(36)
    lens = (64, 64, 1, 200)
    token_lists = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in lens]
    got = enc.embed_token_lists(token_lists).cpu()
    want = _hf_embed(model, cfg, token_lists, "last")
    diff = (got - want).abs()
    cos = (got * want).sum(-1)
    per_seq = diff.pow(2).mean(dim=1).sqrt()
    print(f"\n36-layer real-width forward vs transformers fp32: max |d| = {diff.max():.2e}, "
          f"rms = {diff.pow(2).mean().sqrt():.2e}, min cos = {cos.min():.6f}; per sequence (len: rms): "
          + ", ".join(f"{n}: {v:.1e}" for n, v in zip(lens, per_seq.tolist())))
    assert torch.isfinite(got).all() and torch.allclose(got.norm(dim=1), torch.ones(len(lens)), atol=1e-5)
    assert diff.pow(2).mean().sqrt() <= 3e-3
    assert diff.max() <= 1.5e-2
    assert cos.min() >= 0.9954
    del model


# ------------------------------------------------------------------------------------------------------
# the layer at 16 / 32 token rows as five launches (csrc/crag_encoder_small.hip)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,n,rows,swiglu,pro", [(2560, 6144, 12, False, True), (2560, 19456, 16, True, True),
                                                  (4096, 2560, 10, False, False), (9728, 2560, 10, False, False)])
@pytest.mark.parametrize("m_rows,m_pad", [(1, 16), (16, 16), (17, 32), (32, 32)])
def test_small_gemm_with_fused_residual_rmsnorm_and_swiglu(gpu, k, n, rows, swiglu, pro, m_rows, m_pad):
    """crag_enc_small_gemm, the four forms the 4B layer uses: tiles of 12 / 16 / 10 / 10 weight rows, the residual
    add + RMSNorm prologue (against crag_enc_rmsnorm: the residual sum bit for bit, the projection against torch on
    the unfused kernel's normalised rows) and the SwiGLU epilogue (against crag_enc_swiglu's arithmetic).  Padding
    rows may hold anything and never reach an output."""
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(k + n + 7 * m_rows)
    x = torch.randn(m_pad, k, generator=g) * 2
    d = torch.randn(m_pad, k, generator=g) * 0.5
    x[m_rows:] = float("nan")
    w = (torch.randn(n, k, generator=g) * 0.02).to(BF)
    nw = _bf(1 + 0.1 * torch.randn(k, generator=g))
    xd, dd, wd = _bf(x), _bf(d), w.to(DEV)
    wsw = ops.skinny_gate_up_weight(wd) if swiglu else ops.small_weight(wd, rows)
    n_out = n // 2 if swiglu else n
    out = torch.full((m_rows + 1, n_out), 7.0, dtype=BF, device=DEV)
    if pro:
        res = torch.full((m_pad + 1, k), 5.0, dtype=BF, device=DEV)
        ops.small_gemm(xd, wsw, out[:m_rows], m_rows, n, rows, swiglu=swiglu, delta=dd, norm_w=nw, res_out=res[:m_pad],
                       eps=1e-6)
        normed = torch.empty(m_rows, k, dtype=BF, device=DEV)
        want_res = torch.empty(m_rows, k, dtype=BF, device=DEV)
        ops.rmsnorm(dd[:m_rows].contiguous(), nw, normed, 1e-6, residual_in=xd[:m_rows].contiguous(), residual_out=want_res)
        assert torch.equal(res[:m_rows], want_res)                  # the residual stream: bit for bit
        assert torch.all(res[m_pad] == 5.0) and torch.all(res[m_rows:m_pad] == 5.0)
        # the fused norm sums the squares in another order: a normalised element may land on the neighbouring bf16
        a = normed.float().cpu()
    else:
        ops.small_gemm(xd, wsw, out[:m_rows], m_rows, n, rows)
        a = xd[:m_rows].float().cpu()
    assert torch.all(out[m_rows] == 7.0)                            # nothing written behind the real rows
    ref = a @ w.float().t()
    if not swiglu:
        got = out[:m_rows].float().cpu()
        scale = ref.abs().max()
        assert torch.allclose(got, ref, atol=float(scale) * 2 ** -6, rtol=1.2e-2), (got - ref).abs().max()
    else:
        want = torch.empty(m_rows, n_out, dtype=BF, device=DEV)
        ops.swiglu(ref.to(BF).to(DEV).contiguous(), want)
        got, want = out[:m_rows].float().cpu(), want.float().cpu()
        assert torch.allclose(got, want, atol=float(want.abs().max()) * 2 ** -5, rtol=4e-2), (got - want).abs().max()


@pytest.mark.parametrize("lens", [[16], [1], [5], [32], [17], [16, 16], [7, 9], [3, 20, 9], [1, 1, 1, 1]])
def test_small_attention_fuses_qk_norm_rope_and_causal_attention(gpu, lens):
    """crag_enc_small_attention (<= 32 packed token rows, one workgroup per q head) against the fp32 reference of
    per-head RMSNorm -> RoPE -> causal softmax attention per sequence, and against the two unfused kernels
    (crag_enc_qk_norm_rope + crag_enc_attention) it replaces on that path."""
    from cadence_rag_amd.encoder import ops
    from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder
    hq, hkv = 32, 8
    g = torch.Generator().manual_seed(sum(lens) * 13 + len(lens))
    t = sum(lens)
    qkv = _bf(torch.randn(t + 32, (hq + 2 * hkv) * 128, generator=g))
    qw, kw = _bf(1 + 0.1 * torch.randn(128, generator=g)), _bf(1 + 0.1 * torch.randn(128, generator=g))
    cfg = Qwen3Config(max_length=64)
    table = Qwen3Encoder._rope_table(cfg).to(DEV)
    batch = PackedBatch.build(lens, DEV)
    before = qkv.clone()
    out = torch.full((t + 1, hq * 128), 9.0, dtype=BF, device=DEV)
    ops.small_attention(qkv[:t], qw, kw, table, batch.positions, out[:t], hq, hkv, 1e-6, 1 / math.sqrt(128))
    assert torch.equal(qkv, before) and torch.all(out[t] == 9.0)
    by_tok = torch.empty_like(out)                                # the table's rows gathered per token: same bits
    ops.small_attention(qkv[:t], qw, kw, table.index_select(0, batch.positions.long()), batch.positions, by_tok[:t], hq,
                        hkv, 1e-6, 1 / math.sqrt(128), by_token=True)
    assert torch.equal(by_tok[:t], out[:t])
    got = out[:t].float().cpu().view(t, hq, 128)
    assert torch.isfinite(got).all()
    pos = batch.positions.cpu()

    def normed(block, w, heads):
        v = block.float().cpu().view(t, heads, 128)
        return _rope_ref(v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float().cpu(), pos, cfg.rope_theta)

    ref = _attn_ref(normed(before[:t, : hq * 128], qw, hq), normed(before[:t, hq * 128: (hq + hkv) * 128], kw, hkv),
                    before[:t, (hq + hkv) * 128:].float().cpu().view(t, hkv, 128), lens, hq, hkv)
    assert torch.allclose(got, ref, atol=3e-2, rtol=3e-2), (lens, (got - ref).abs().max())
    # the kernels it replaces
    vt = torch.empty(hkv, 128, batch.t_pad, dtype=BF, device=DEV)
    ops.qk_rope_vt(qkv, qw, kw, table, batch.positions, hq, hkv, 1e-6, vt, batch.tok_of_pad)
    old = torch.empty(t, hq * 128, dtype=BF, device=DEV)
    ops.attention(qkv, vt, old, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0, hq, hkv, 1 / math.sqrt(128))
    assert torch.allclose(got, old.float().cpu().view(t, hq, 128), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("lens", [[16] * 8, [16, 3, 9, 16, 1, 12, 7, 15], [32, 17, 5, 32], [1], [9, 32], [2] * 70])
def test_small_attention_over_a_packed_batch_of_short_sequences(gpu, lens):
    """crag_enc_small_attention_seqs: one workgroup per (q head, sequence) over a packed batch whose sequences all have
    <= 32 tokens (ragged starts: the blocks follow cu_seqlens) -- against the fp32 reference, against the two launches
    it replaces (crag_enc_qk_rope_vt + crag_enc_attention), and bit for bit against crag_enc_small_attention run on each
    sequence by itself (the same kernel, one block)."""
    from cadence_rag_amd.encoder import ops
    from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder
    hq, hkv = 32, 8
    g = torch.Generator().manual_seed(sum(lens) * 7 + len(lens))
    t = sum(lens)
    qkv = _bf(torch.randn(t + 32, (hq + 2 * hkv) * 128, generator=g))
    qw, kw = _bf(1 + 0.1 * torch.randn(128, generator=g)), _bf(1 + 0.1 * torch.randn(128, generator=g))
    cfg = Qwen3Config(max_length=64)
    table = Qwen3Encoder._rope_table(cfg).to(DEV)
    batch = PackedBatch.build(lens, DEV)
    assert batch.max_len == max(lens)
    before = qkv.clone()
    out = torch.full((t + 1, hq * 128), 9.0, dtype=BF, device=DEV)
    cs_tok = table.index_select(0, batch.positions.long())
    ops.small_attention_seqs(qkv[:t], qw, kw, cs_tok, batch.positions, batch.cu, len(lens), batch.max_len, out[:t], hq, hkv,
                             1e-6, 1 / math.sqrt(128), by_token=True)
    assert torch.equal(qkv, before) and torch.all(out[t] == 9.0)
    full_table = torch.full_like(out, 7.0)                        # the whole table + positions: same bits
    ops.small_attention_seqs(qkv[:t], qw, kw, table, batch.positions, batch.cu, len(lens), batch.max_len, full_table[:t],
                             hq, hkv, 1e-6, 1 / math.sqrt(128))
    assert torch.equal(full_table[:t], out[:t])
    got = out[:t].float().cpu().view(t, hq, 128)
    assert torch.isfinite(got).all()
    pos = batch.positions.cpu()

    def normed(block, w, heads):
        v = block.float().cpu().view(t, heads, 128)
        return _rope_ref(v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float().cpu(), pos, cfg.rope_theta)

    ref = _attn_ref(normed(before[:t, : hq * 128], qw, hq), normed(before[:t, hq * 128: (hq + hkv) * 128], kw, hkv),
                    before[:t, (hq + hkv) * 128:].float().cpu().view(t, hkv, 128), lens, hq, hkv)
    assert torch.allclose(got, ref, atol=3e-2, rtol=3e-2), (lens, (got - ref).abs().max())
    vt = torch.empty(hkv, 128, batch.t_pad, dtype=BF, device=DEV)
    work = qkv.clone()
    ops.qk_rope_vt(work, qw, kw, table, batch.positions, hq, hkv, 1e-6, vt, batch.tok_of_pad)
    old = torch.empty(t, hq * 128, dtype=BF, device=DEV)
    ops.attention(work, vt, old, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0, hq, hkv, 1 / math.sqrt(128))
    assert torch.allclose(got, old.float().cpu().view(t, hq, 128), atol=2e-2, rtol=2e-2)
    # the same projection handed over as two split-K partial tiles (fp32, token-major): summed and rounded while loaded
    m_pad = (t + 31) // 32 * 32
    a = torch.randn(m_pad, qkv.shape[1], generator=g).to(DEV) * 0.5
    parts = torch.zeros(2, m_pad, qkv.shape[1], dtype=torch.float32, device=DEV)
    parts[0, :t], parts[1, :t] = a[:t], qkv[:t].float() - a[:t]   # (fp32 sums that round back to the bf16 values)
    rounded = (parts[0, :t] + parts[1, :t]).to(BF)
    from_parts = torch.full_like(out, 5.0)
    ops.small_attention_seqs_parts(parts, 2, m_pad, qw, kw, cs_tok, batch.positions, batch.cu, len(lens), batch.max_len,
                                   from_parts[:t], hq, hkv, 1e-6, 1 / math.sqrt(128), by_token=True)
    direct = torch.full_like(out, 5.0)
    ops.small_attention_seqs(rounded, qw, kw, cs_tok, batch.positions, batch.cu, len(lens), batch.max_len, direct[:t], hq, hkv,
                             1e-6, 1 / math.sqrt(128), by_token=True)
    assert torch.equal(from_parts, direct)
    # each sequence by itself through the one-block entry point (the block size is picked per call there: compare the
    # sequences whose own length picks the same kernel as the batch's longest)
    lo = 0
    for n in lens[:6]:
        if (n <= 16) == (batch.max_len <= 16):
            one = torch.empty(n, hq * 128, dtype=BF, device=DEV)
            ops.small_attention(qkv[lo:lo + n], qw, kw, table, batch.positions[lo:lo + n].contiguous(), one, hq, hkv, 1e-6,
                                1 / math.sqrt(128))
            assert torch.equal(one, out[lo:lo + n]), (lens, lo)
        lo += n


@pytest.mark.parametrize("k,n,swiglu,splitk", [(2560, 6144, False, 4), (4096, 2560, False, 8), (9728, 2560, False, 8),
                                              (9728, 2560, False, 1), (2560, 19456, True, 1), (2560, 19456, True, 3)])
@pytest.mark.parametrize("m_rows,m_pad", [(128, 128), (97, 128), (96, 96), (70, 96), (64, 64), (33, 64), (32, 32), (17, 32)])
def test_wide_gemm_streams_the_weights_at_64_and_128_rows(gpu, k, n, swiglu, splitk, m_rows, m_pad):
    """crag_enc_wide_gemm + crag_enc_wide_reduce (the gateway's batch sizes, RUNBOOK:304,331-334: up to 8 short queries
    = 128 token rows) against torch: bf16 operands, fp32 accumulation over K splits added in split order, one rounding;
    the SwiGLU form against crag_enc_swiglu's arithmetic on the bf16-rounded reference projection; uneven K splits
    (9728 / 128 = 76 chunks over 8 workgroups); rows beyond m_rows are never written."""
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(k + n + m_rows + splitk)
    x = torch.randn(m_pad, k, generator=g).to(BF)
    x[m_rows:] = 0.0                                            # padding rows are read: finite values
    w = (torch.randn(n, k, generator=g) * 0.02).to(BF)
    xd, wd = x.to(DEV), w.to(DEV)
    ref = x[:m_rows].float() @ w.float().t()
    width = n // 2 if swiglu else n
    out = torch.full((m_pad + 1, width), 7.0, dtype=BF, device=DEV)
    ww = ops.wide_gate_up_weight(wd) if swiglu else ops.wide_weight(wd)
    ops.wide_gemm(xd, ww, out, m_rows, n, splitk, swiglu=swiglu)
    assert torch.all(out[m_rows:] == 7.0)                       # nothing written behind the real rows
    got = out[:m_rows].float().cpu()
    if not swiglu:
        assert torch.allclose(got, ref, atol=2e-2, rtol=1.2e-2), (got - ref).abs().max()
        assert (got - ref.to(BF).float()).abs().max() <= 2 * ref.abs().max() * 2 ** -8
    else:
        want = torch.empty(m_rows, width, dtype=BF, device=DEV)
        ops.swiglu(ref.to(BF).to(DEV).contiguous(), want)
        assert torch.allclose(got, want.float().cpu(), atol=3e-3, rtol=3e-2), (got - want.float().cpu()).abs().max()


@pytest.mark.parametrize("m_rows,m_pad,splitk", [(32, 32, 4), (17, 32, 4), (64, 64, 4), (50, 64, 2), (96, 96, 8), (80, 96, 4),
                                                 (128, 128, 8), (99, 128, 8)])
def test_split_k_partials_summed_by_the_norm_that_consumes_them(gpu, m_rows, m_pad, splitk):
    """crag_enc_wide_gemm_rows (the split-K partial tiles token-major) + crag_enc_rmsnorm_partials (sum of the splits,
    residual add, RMSNorm in ONE launch) against wide_gemm (partials + reduce launch) + rmsnorm: the same bits for the
    normed rows and for the new residual stream -- the down projection -> next layer's ln1 hand-off of the 32 / 64 /
    128-row forward."""
    from cadence_rag_amd.encoder import ops
    g = torch.Generator().manual_seed(m_rows * 7 + splitk)
    k, n = 9728, 2560
    w = _bf(torch.randn(n, k, generator=g) * 0.02)
    ww = ops.wide_weight(w)
    x = _bf(torch.randn(m_pad, k, generator=g))
    resid = _bf(torch.randn(m_rows, n, generator=g))
    nw = _bf(1 + 0.1 * torch.randn(n, generator=g))
    delta = torch.empty(m_rows, n, dtype=BF, device=DEV)
    ops.wide_gemm(x, ww, delta, m_rows, n, splitk)
    want_norm, want_res = torch.empty_like(delta), torch.empty_like(delta)
    ops.rmsnorm(delta, nw, want_norm, 1e-6, residual_in=resid, residual_out=want_res)
    scratch = torch.full((splitk * n * m_pad + 64,), float("nan"), dtype=torch.float32, device=DEV)
    parts = ops.wide_gemm_rows(x, ww, n, splitk, scratch=scratch)
    got_norm = torch.full((m_rows + 1, n), 9.0, dtype=BF, device=DEV)
    res_io = resid.clone()                                        # residual_out aliases residual_in, as the forward does
    ops.rmsnorm_partials(parts, splitk, m_pad, nw, got_norm[:m_rows], 1e-6, residual_in=res_io, residual_out=res_io)
    torch.cuda.synchronize()
    assert torch.all(torch.isnan(scratch[splitk * n * m_pad:])) and torch.all(got_norm[m_rows] == 9.0)
    assert torch.equal(res_io, want_res)
    assert torch.equal(got_norm[:m_rows], want_norm)


@pytest.mark.parametrize("lens", [[16, 3, 9, 16, 1, 12, 7, 15], [16, 2, 11, 5], [40], [100], [30, 32, 17, 9],
                                  [16, 5, 9], [16, 3, 9, 16, 1, 12, 7],    # (these two: rounded up with phantom sequences)
                                  [20], [32], [5, 16],                       # (32 rows: the wide kernels at 32 rows)
                                  [16, 9, 16, 2, 11], [16] * 6, [70], [32, 20, 30]])   # (96 rows, the first after rounding up)
def test_three_to_eight_short_queries_use_the_wide_projections_and_match_transformers(gpu, monkeypatch, lens):
    """The gateway's batch sizes (max_batch_size 8, preferred 1 / 2 / 4 / 8, RUNBOOK:304,331-334): 3 to 8 queries of
    <= 16 tokens, or one query of 33 to 128 tokens = one graph replay over 64 or 128 token rows, whose gate|up + SwiGLU
    (and, at 128 rows, down) are the weight-streaming kernels of crag_encoder_wide.hip.  Real 4B widths, two layers;
    against transformers' Qwen3Model in fp32 and against the same forward through the library GEMMs
    (CRAG_ENC_NO_WIDE=1)."""
    model, enc, cfg = _real_width_hf_and_mine()
    rng = np.random.default_rng(8)
    token_lists = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in lens]
    monkeypatch.delenv("CRAG_ENC_NO_GRAPH", raising=False)
    monkeypatch.delenv("CRAG_ENC_NO_WIDE", raising=False)
    fast = enc.embed_token_lists(token_lists)
    assert enc.__dict__.get("_wide") is not None and not enc.__dict__.get("_wide_off", False)
    assert torch.equal(fast, enc.embed_token_lists(token_lists))
    monkeypatch.setenv("CRAG_ENC_NO_WIDE", "1")
    lib = enc.embed_token_lists(token_lists)     # (the switch drops the graphs captured over the wide kernels)
    assert enc.__dict__.get("_wide_off") is True
    want = _hf_embed(model, cfg, token_lists, "last")
    diff = (fast.cpu() - want).abs()
    assert torch.allclose(fast.norm(dim=1).cpu(), torch.ones(len(lens)), atol=1e-5)
    assert diff.pow(2).mean().sqrt() <= 6.5e-4 and diff.max() <= 3e-3
    assert ((fast.cpu() * want).sum(-1)).min() >= 0.9998
    assert (fast - lib).abs().max() <= 3e-3


def test_one_short_query_takes_the_five_launch_layer_and_matches_transformers(gpu, monkeypatch):
    """Real 4B widths, two layers, the shapes of the reference's operating point (one query of <= 16 tokens; also one
    of <= 32 and two of <= 16): embed_token_lists replays the graph of the five-launch layer
    (crag_encoder_small.hip).  Against transformers' Qwen3Model in fp32 (the bars of the real-width test above) and
    against the eager packed forward through the library GEMMs and the unfused kernels."""
    model, enc, cfg = _real_width_hf_and_mine()
    rng = np.random.default_rng(5)
    monkeypatch.setenv("CRAG_ENC_NO_WIDE_32", "1")   # 32 rows through the five-launch layer here (default: the wide kernels)
    for lens in ([9], [16], [1], [20], [32], [5, 16], [12, 3]):
        token_lists = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in lens]
        monkeypatch.delenv("CRAG_ENC_NO_GRAPH", raising=False)
        monkeypatch.delenv("CRAG_ENC_NO_SKINNY", raising=False)
        fast = enc.embed_token_lists(token_lists)
        assert torch.equal(fast, enc.embed_token_lists(token_lists))
        assert enc._skinny is not None and not enc._skinny_v1
        monkeypatch.setenv("CRAG_ENC_NO_GRAPH", "1")
        monkeypatch.setenv("CRAG_ENC_NO_SKINNY", "1")
        eager = enc.embed_token_lists(token_lists)
        want = _hf_embed(model, cfg, token_lists, "last")
        diff = (fast.cpu() - want).abs()
        print(f"\nfive-launch layer {lens}: vs transformers max |d| = {diff.max():.2e}, rms = "
              f"{diff.pow(2).mean().sqrt():.2e}; vs eager max |d| = {(fast - eager).abs().max():.2e}")
        assert torch.allclose(fast.norm(dim=1).cpu(), torch.ones(len(lens)), atol=1e-5)
        assert diff.pow(2).mean().sqrt() <= 6.5e-4 and diff.max() <= 3e-3, lens
        assert ((fast.cpu() * want).sum(-1)).min() >= 0.9998, lens
        assert (fast - eager).abs().max() <= 3e-3, lens
    # mean pooling takes no graph, but 16 / 32 token rows still run the five-launch layer (its final RMSNorm + mean)
    monkeypatch.delenv("CRAG_ENC_NO_GRAPH", raising=False)
    monkeypatch.delenv("CRAG_ENC_NO_SKINNY", raising=False)
    enc.cfg.pooling = "mean"
    try:
        for lens in ([16], [7, 9], [32]):
            token_lists = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in lens]
            got = enc.embed_token_lists(token_lists).cpu()
            want = _hf_embed(model, cfg, token_lists, "mean")
            diff = (got - want).abs()
            assert diff.pow(2).mean().sqrt() <= 6.5e-4 and diff.max() <= 3e-3, (lens, diff.max())
    finally:
        enc.cfg.pooling = "last"


def test_backfill_through_native_encoder_into_hbm_index(gpu, monkeypatch):
    """BASELINE configs[3] shape on one GPU: run_embedding_backfill (embedding_pipeline.py:241) at batch 256 over
    chunks of ~256 tokens, through embed_texts -> the native encoder at the 4B widths (4 layers: the per-layer
    operators and shapes are the bench's; depth only repeats them) -> the store -> a DenseTable sink in HBM;
    then the index is searched with embeddings of the same texts."""
    from uuid import UUID
    from cadence_rag_amd import embedding_pipeline as ep, embeddings, retrieve as rt
    from cadence_rag_amd.config import settings
    from cadence_rag_amd.encoder.qwen3 import ByteTokenizer, Qwen3Config, Qwen3Encoder
    cfg = Qwen3Config(num_layers=4, vocab_size=4096)
    enc = Qwen3Encoder.random_init(cfg, seed=99, device=DEV)
    enc.tokenizer = ByteTokenizer(eos_id=256)
    rng = np.random.default_rng(5)
    words = ["alpha", "gateway", "ECONNRESET", "retry", "ticket", "ABC-123", "v1.2.3", "timeout", "the", "of", "api"]

    def text(i):  # ~256 bytes (= tokens) on average, 40 .. 700
        target = int(np.clip(rng.normal(255, 90), 40, 700))
        out = f"chunk {i}:"
        while len(out) < target:
            out += " " + words[int(rng.integers(len(words)))]
        return out[:target]

    n_chunks, n_art = 600, 150
    tables = {"chunks": {1000 + i: {"call_id": UUID(int=1 + i % 5), "text": text(i), "embedding": None}
                         for i in range(n_chunks)},
              "artifact_chunks": {7 + i: {"call_id": UUID(int=1 + i % 5), "text": text(i), "embedding": None}
                                  for i in range(n_art)}}
    tables["chunks"][1003]["text"] = "   "          # blank rows are never fetched (embedding_pipeline.py:134-135)
    chunk_table = rt.DenseTable("chunks", "chunk_id", dim=1024, capacity=256)   # grows while the backfill runs
    art_index = DenseIndex(1024, capacity=n_art)

    def chunk_columns(ids):
        return {"chunk_id": list(ids), "call_id": [tables["chunks"][i]["call_id"] for i in ids],
                "speaker": ["S"] * len(ids), "start_ts_ms": [0] * len(ids), "end_ts_ms": [1] * len(ids),
                "text": [tables["chunks"][i]["text"] for i in ids]}

    store = ep.InMemoryStore(tables, sinks={"chunks": chunk_table.sink(chunk_columns), "artifact_chunks": art_index})
    monkeypatch.setattr(settings, "embeddings_base_url", "native")
    monkeypatch.setattr(settings, "embeddings_dim", 1024)
    embeddings.set_encoder(enc)
    ep.set_store(store)
    seen_batches = []
    real_encode = enc.encode
    monkeypatch.setattr(enc, "encode", lambda texts: (seen_batches.append(len(texts)), real_encode(texts))[1])
    try:
        summary = ep.run_embedding_backfill(batch_size=256)
        assert summary.rows_updated == n_chunks - 1 + n_art and summary.per_table == {
            "chunks": n_chunks - 1, "artifact_chunks": n_art}
        assert summary.calls_touched == 5 and summary.model_used == cfg.model_id
        assert seen_batches == [256, 256, 87, 150]                  # batch 256, last batch ragged, then table 2
        assert len(chunk_table) == n_chunks - 1 and len(art_index) == n_art
        assert tables["chunks"][1003]["embedding"] is None
        # what sits in HBM is bit for bit what the store recorded
        rows, ids = chunk_table.index.get_rows(0, len(chunk_table))
        assert ids.tolist() == [i for i in sorted(tables["chunks"]) if i != 1003]
        stored = np.array([tables["chunks"][i]["embedding"] for i in ids], dtype=np.float32)
        assert np.array_equal(rows, stored)
        assert np.allclose(np.linalg.norm(rows, axis=1), 1.0, atol=1e-5)
        # search the freshly filled index: re-embedding a stored text finds its own row first (the batch it
        # is embedded in differs, so the score is 1 up to bf16 GEMM tiling effects), and the scan agrees
        # with the oracle over the stored vectors
        probe = [1000, 1100, 1255, 1256, 1599]
        res = embeddings.embed_texts([tables["chunks"][i]["text"] for i in probe])
        got = chunk_table.index.search(np.array(res.vectors, dtype=np.float32), 10)
        assert got[0][:, 0].tolist() == probe and np.all(got[1][:, 0] > 0.999)
        want = oracle_topk(np.array(res.vectors, dtype=np.float32), rows, 10, ids)
        from tests.helpers import assert_topk_matches
        assert_topk_matches(*got, *want, tol=1e-4)
        hits = rt._fetch_chunks_dense(chunk_table, res.vectors[1], None, None, "exact", 3)
        assert hits[0]["chunk_id"] == 1100 and hits[0]["text"] == tables["chunks"][1100]["text"]
        # a second run finds nothing left to do
        again = ep.run_embedding_backfill(batch_size=256)
        assert again.rows_updated == 0
        # the device-resident form of the same job: embed_texts_device -> one CUDA tensor per batch -> the sink's
        # crag_index_add with a device pointer; no Python float is ever made.  Same batches, same kernels: the
        # rows in HBM equal the host-list run's.
        tables2 = {name: {i: dict(r, embedding=None) for i, r in t.items()} for name, t in tables.items()}
        dev_chunks, dev_arts = DenseIndex(1024, capacity=n_chunks), DenseIndex(1024, capacity=n_art)
        try:
            ep.set_store(ep.DeviceSinkStore(tables2, sinks={"chunks": dev_chunks, "artifact_chunks": dev_arts}))
            seen_batches.clear()
            s2 = ep.run_embedding_backfill(batch_size=256)
            assert s2.rows_updated == summary.rows_updated and s2.per_table == summary.per_table
            assert seen_batches == []      # enc.encode (the list-returning form) is not on this path
            rows2, ids2 = dev_chunks.get_rows(0, len(dev_chunks))
            assert np.array_equal(ids2, ids) and np.allclose(rows2, rows, atol=1e-6)
            assert all(r["embedding"] == "hbm" for i, r in tables2["chunks"].items() if i != 1003)
        finally:
            dev_chunks.close()
            dev_arts.close()
    finally:
        embeddings.set_encoder(None)
        ep.set_store(None)
        chunk_table.close()
        art_index.close()


def oracle_topk(q, corpus, k, ids):
    import oracle
    return oracle.exact_topk(q, corpus, k, ids=ids, mode=oracle.F64)
