"""Builds a tiny LOCAL Hugging Face checkpoint directory — config.json, model.safetensors, tokenizer.json and the
tokenizer config files — so that the real loading path of the encoder (Qwen3Encoder.from_pretrained: AutoTokenizer,
safetensors, tensor-name mapping, truncation through the tokenizer) runs in tests without any download.

Everything is this repository's own data: a byte-level BPE tokenizer trained here on the sentences below with the
`tokenizers` library, and a seeded random 2-layer Qwen3 model written by `transformers` itself.  Shape of the real
thing (P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:657-660, 689-699): a fast tokenizer whose post-processor appends the
end-of-text token, pad = eos, loaded with local_files_only=True.
"""
from __future__ import annotations

from pathlib import Path

import torch

EOS = "<|endoftext|>"
CORPUS = [
    "the customer called about a failed deployment of the api gateway",
    "we saw ECONNRESET errors between the gateway and the billing service after the upgrade",
    "ticket ABC-123 tracks the rollback to version v1.2.3 and the follow up call next week",
    "the agent confirmed the refund and scheduled a call back for tuesday morning",
    "latency went from forty milliseconds to nine hundred during the incident window",
    "please send the transcript and the action items to the account team",
    "the speaker asked whether the embedding backfill had finished for all chunks",
    "numbers 0 1 2 3 4 5 6 7 8 9 and punctuation , . ; : ! ? ( ) [ ] { } - _ / \\ ' \"",
]


def build_tokenizer(vocab_size: int = 384):
    """Byte-level BPE (every string is encodable) + TemplateProcessing that appends EOS, as Qwen's tokenizer does."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=[EOS],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(CORPUS * 4, trainer)
    eos_id = tok.token_to_id(EOS)
    tok.post_processor = processors.TemplateProcessing(single=f"$A {EOS}", pair=f"$A {EOS} $B:1 {EOS}:1",
                                                       special_tokens=[(EOS, eos_id)])
    return PreTrainedTokenizerFast(tokenizer_object=tok, eos_token=EOS, pad_token=EOS)


def build_checkpoint(root, *, prefix: str = "", layers: int = 2, seed: int = 4242):
    """Writes the directory and returns (hf_model fp32 on the CPU with bf16-representable weights, tokenizer).
    prefix "model.": tensor names as a *ForCausalLM checkpoint stores them (+ an lm_head the encoder must ignore)."""
    from safetensors.torch import save_file
    from transformers import Qwen3Config as HFConfig
    from transformers.models.qwen3.modeling_qwen3 import Qwen3Model
    root = Path(root)
    root.mkdir(parents=True, exist_ok=True)
    tokenizer = build_tokenizer()
    tokenizer.save_pretrained(str(root))
    torch.manual_seed(seed)
    cfg = HFConfig(vocab_size=len(tokenizer), hidden_size=256, intermediate_size=512, num_hidden_layers=layers,
                   num_attention_heads=4, num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-6,
                   max_position_embeddings=2048, rope_parameters={"rope_theta": 1_000_000.0, "rope_type": "default"},
                   attention_bias=False, tie_word_embeddings=False)
    model = Qwen3Model(cfg).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(1 + 0.1 * torch.randn_like(p) if "norm" in name else torch.randn_like(p) * 0.05)
            p.copy_(p.to(torch.bfloat16).float())   # the encoder holds bf16 weights: same values on both sides
    cfg.save_pretrained(str(root))                   # config.json
    sd = {prefix + k: v.to(torch.bfloat16).contiguous() for k, v in model.state_dict().items()}
    if prefix:
        sd["lm_head.weight"] = torch.zeros(len(tokenizer), cfg.hidden_size, dtype=torch.bfloat16)
    save_file(sd, str(root / "model.safetensors"))
    return model, tokenizer
