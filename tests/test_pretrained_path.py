"""E1 made real: the encoder's checkpoint-loading path — Qwen3Encoder.from_pretrained = AutoTokenizer
(local_files_only) + safetensors + the transformers tensor-name mapping — and truncation THROUGH the tokenizer, as the
gateway does it (P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:657-660 startup, :689-699 tokenizer call, :703-715 pooling).
The checkpoint directory is built on the spot by tests/tiny_checkpoint.py from this repository's own data (a byte-level
BPE tokenizer trained there, a seeded random 2-layer Qwen3 written by transformers): nothing is downloaded.
"""
import numpy as np
import pytest
import torch

from tests.tiny_checkpoint import EOS, build_checkpoint

LONG = "the gateway returned ECONNRESET again and again " * 400     # far more than 1024 tokens
TEXTS = ["  where did we discuss ECONNRESET in the api gateway?  ", "", "ticket ABC-123 v1.2.3", "   ",
         LONG, "naïve café — ünïcödé ✓", "a"]


@pytest.mark.parametrize("prefix", ["", "model."])
def test_from_pretrained_maps_names_and_loads_the_tokenizer(tmp_path, prefix):
    from cadence_rag_amd.encoder.qwen3 import Qwen3Encoder
    model, tok = build_checkpoint(tmp_path / "ckpt", prefix=prefix)
    enc = Qwen3Encoder.from_pretrained(str(tmp_path / "ckpt"), device=torch.device("cpu"), out_dim=64)
    c = enc.cfg
    assert (c.hidden_size, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim, c.intermediate_size) == \
           (256, 2, 4, 2, 128, 512)
    assert c.vocab_size == len(tok) and c.rope_theta == 1_000_000.0 and c.max_length == 1024 and c.out_dim == 64
    sd = model.state_dict()
    for i, layer in enumerate(enc.layers):
        p = f"layers.{i}."
        want_qkv = torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"],
                              sd[p + "self_attn.v_proj.weight"]]).to(torch.bfloat16)
        assert torch.equal(layer["qkv"], want_qkv)
        assert torch.equal(layer["gate_up"], torch.cat([sd[p + "mlp.gate_proj.weight"],
                                                        sd[p + "mlp.up_proj.weight"]]).to(torch.bfloat16))
        assert torch.equal(layer["o"], sd[p + "self_attn.o_proj.weight"].to(torch.bfloat16))
        assert torch.equal(layer["down"], sd[p + "mlp.down_proj.weight"].to(torch.bfloat16))
        assert torch.equal(layer["q_norm"], sd[p + "self_attn.q_norm.weight"].to(torch.bfloat16))
        assert torch.equal(layer["ln2"], sd[p + "post_attention_layernorm.weight"].to(torch.bfloat16))
    assert torch.equal(enc.embed, sd["embed_tokens.weight"].to(torch.bfloat16))
    assert torch.equal(enc.final_norm, sd["norm.weight"].to(torch.bfloat16))
    # the tokenizer is the directory's own fast tokenizer; truncation happens inside it and keeps the end marker
    assert enc.tokenizer.is_fast and enc.tokenizer.eos_token == EOS
    ids = enc.tokenize([LONG, "a", "ticket ABC-123"])
    eos = enc.tokenizer.eos_token_id
    assert len(ids[0]) == 1024 and ids[0][-1] == eos and ids[1][-1] == eos and len(ids[1]) == 2
    assert ids[0] == tok(LONG, truncation=True, max_length=1024)["input_ids"]
    assert enc.tokenizer.decode(ids[2][:-1]) == "ticket ABC-123"


def test_from_pretrained_reports_missing_pieces(tmp_path):
    from cadence_rag_amd.encoder.qwen3 import Qwen3Encoder
    build_checkpoint(tmp_path / "ckpt")
    (tmp_path / "ckpt" / "model.safetensors").unlink()
    with pytest.raises(FileNotFoundError, match="safetensors"):
        Qwen3Encoder.from_pretrained(str(tmp_path / "ckpt"), device=torch.device("cpu"), out_dim=64)
    build_checkpoint(tmp_path / "ckpt2")
    with pytest.raises(ValueError, match="out_dim"):
        Qwen3Encoder.from_pretrained(str(tmp_path / "ckpt2"), device=torch.device("cpu"), out_dim=512)


@pytest.mark.gpu
@pytest.mark.parametrize("prefix", ["", "model."])
def test_embed_texts_through_a_loaded_checkpoint_matches_transformers(gpu, tmp_path, monkeypatch, prefix):
    """texts -> embed_texts (blank filtering, strip: embeddings.py:29-33) -> the directory's tokenizer (truncation at
    1024) -> packed forward on the GPU -> last-token pool -> [:64] -> L2, against transformers' Qwen3Model on the CPU
    fed by the same tokenizer one text at a time."""
    from cadence_rag_amd import embeddings
    from cadence_rag_amd.config import settings
    from cadence_rag_amd.encoder.qwen3 import Qwen3Encoder
    model, tok = build_checkpoint(tmp_path / "ckpt", prefix=prefix)
    enc = Qwen3Encoder.from_pretrained(str(tmp_path / "ckpt"), device=torch.device("cuda", 0), out_dim=64,
                                       model_id="tiny-qwen3-test")
    monkeypatch.setattr(settings, "embeddings_base_url", "native")
    monkeypatch.setattr(settings, "embeddings_dim", 64)
    embeddings.set_encoder(enc)
    try:
        res = embeddings.embed_texts(TEXTS)
        kept = [t.strip() for t in TEXTS if t.strip()]
        assert len(res.vectors) == len(kept) == 5 and res.model == "tiny-qwen3-test"
        want = []
        with torch.no_grad():
            for t in kept:
                ids = tok(t, truncation=True, max_length=1024)["input_ids"]
                h = model(input_ids=torch.tensor([ids])).last_hidden_state[0, -1, :64].float()
                want.append(h / h.norm().clamp_min(1e-12))
        want = torch.stack(want)
        got = torch.tensor(res.vectors)
        diff = (got - want).abs()
        cos = (got * want).sum(-1)
        print(f"\nloaded checkpoint ({prefix or 'no'} prefix): max |d| = {diff.max():.2e}, rms = "
              f"{diff.pow(2).mean().sqrt():.2e}, min cos = {cos.min():.6f}")
        # 2 layers of bf16 (12 roundings each) on a unit 64-d output: the toy-forward bars of test_encoder_gpu.py
        assert cos.min() > 0.9995 and diff.pow(2).mean().sqrt() < 2.5e-3 and diff.max() < 1e-2
        assert torch.allclose(got.norm(dim=1), torch.ones(5), atol=1e-4)
        # batching changes nothing but the GEMM tiling
        parts = embeddings.embed_texts_batched(kept, batch_size=2)
        assert np.allclose(np.array(parts.vectors), np.array(res.vectors), atol=1e-2)
        with pytest.raises(embeddings.EmbeddingClientError, match="at least one non-empty text"):
            embeddings.embed_texts(["", "   "])
    finally:
        embeddings.set_encoder(None)
