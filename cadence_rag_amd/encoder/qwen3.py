"""Qwen3-Embedding encoder on MI355X.

Forward of one packed batch (T tokens of B sequences, no padding):
  embed_gather -> 36 x [ rmsnorm(+residual) -> QKV GEMM -> q/k head-norm + RoPE (in place) ->
  V transpose -> causal GQA flash attention -> O GEMM -> rmsnorm(+residual) -> gate|up GEMM ->
  SwiGLU -> down GEMM ] -> last-token pool + final RMSNorm + [:out_dim] + fp32 L2 normalise.
The GEMMs are plain library GEMMs (torch.nn.functional.linear -> hipBLASLt); every other
operator is hand-written HIP (csrc/crag_encoder.hip).

Replaces: the external gateway behind /root/reference/app/embeddings.py:53-59 whose math is
documented in P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:683-716 (tokenise, truncate to 1024,
forward, LAST-token pooling, slice 2560 -> 1024, L2 normalise with max(norm, 1e-12)).  Packed
sequences mean the gateway's left-padding pooling quirk cannot occur: the pooled
position is always the last real token.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


@dataclass
class Qwen3Config:
    """Architecture of Qwen/Qwen3-Embedding-4B (public model family; verify against config.json
    when real weights are loaded)."""
    hidden_size: int = 2560
    num_layers: int = 36
    num_heads: int = 32
    num_kv_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 9728
    vocab_size: int = 151665
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    max_length: int = 1024      # gateway truncation (RUNBOOK:484,748)
    out_dim: int = 1024         # EMBEDDINGS_DIM: slice of the 2560-d state (RUNBOOK:712-713)
    pooling: str = "last"       # "last" (reference gateway) or "mean"
    model_id: str = "Qwen/Qwen3-Embedding-4B"

    @property
    def q_size(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_kv_heads * self.head_dim

    def flops_per_token(self, avg_context: float) -> float:
        """Forward FLOPs per token: 2 * non-embedding params + causal attention."""
        per_layer = self.hidden_size * (self.q_size + 2 * self.kv_size) + self.q_size * self.hidden_size \
            + 3 * self.hidden_size * self.intermediate_size
        attn = 4.0 * (avg_context / 2.0) * self.head_dim * self.num_heads
        return self.num_layers * (2.0 * per_layer + attn)

    def flops_skipped_in_last_layer(self, n_tokens: int, n_seqs: int) -> float:
        """With last-token pooling the last layer's output projection and MLP run on the pooled rows only
        (Qwen3Encoder.forward_packed): FLOPs of the full-model count that are NOT executed."""
        if self.pooling != "last":
            return 0.0
        return 2.0 * (self.q_size * self.hidden_size + 3 * self.hidden_size * self.intermediate_size) * max(n_tokens - n_seqs, 0)


@dataclass
class PackedBatch:
    """Device-side description of a packed batch (built once per batch on the host)."""
    n_tokens: int
    n_seqs: int
    t_pad: int
    cu: torch.Tensor          # int32 [B+1]
    cu_pad: torch.Tensor      # int32 [B+1], 32-aligned starts in the transposed-V key axis
    positions: torch.Tensor   # int32 [T]
    tok_of_pad: torch.Tensor  # int32 [t_pad]
    blk_seq: torch.Tensor     # int32 [n_blocks]
    blk_q0: torch.Tensor      # int32 [n_blocks]
    last_tok: torch.Tensor    # int64 [B]: packed position of every sequence's last token
    cu_one: torch.Tensor      # int32 [B+1] = 0..B: the batch of last tokens as B one-token sequences
    max_len: int = 0          # longest sequence (host side: picks the attention kernel)
    cs_tok: Optional[torch.Tensor] = None   # the RoPE table's rows of `positions`, [T, 64, 2] fp32: set by a caller that
                                            # reuses the batch (the graph path: positions are a constant of the shape)

    @staticmethod
    def build(lengths: Sequence[int], device) -> "PackedBatch":
        lens = np.asarray(lengths, dtype=np.int64)
        if lens.size == 0 or (lens <= 0).any():
            raise ValueError("every sequence needs at least one token")
        cu = np.zeros(lens.size + 1, dtype=np.int64)
        np.cumsum(lens, out=cu[1:])
        padded = (lens + 31) // 32 * 32
        cu_pad = np.zeros(lens.size + 1, dtype=np.int64)
        np.cumsum(padded, out=cu_pad[1:])
        t, t_pad = int(cu[-1]), int(cu_pad[-1])
        seq_of_tok = np.repeat(np.arange(lens.size), lens)
        positions = np.arange(t) - cu[seq_of_tok]
        tok_of_pad = np.full(t_pad, -1, dtype=np.int64)
        tok_of_pad[cu_pad[seq_of_tok] + positions] = np.arange(t)
        nblk = padded // 32
        blk_seq = np.repeat(np.arange(lens.size), nblk)
        blk_q0 = (np.arange(int(nblk.sum())) - np.repeat(np.cumsum(nblk) - nblk, nblk)) * 32
        # longest blocks first: the last q block of a long sequence walks the most key tiles
        order = np.argsort(-blk_q0, kind="stable")

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device, non_blocking=True)

        last_tok = torch.from_numpy(np.ascontiguousarray(cu[1:] - 1)).to(device, non_blocking=True)
        return PackedBatch(t, int(lens.size), t_pad, dev(cu), dev(cu_pad), dev(positions), dev(tok_of_pad),
                           dev(blk_seq[order]), dev(blk_q0[order]), last_tok, dev(np.arange(lens.size + 1)),
                           int(lens.max()))


QKV_ROW_CHUNK = 32768


class Qwen3Encoder:
    """Weights live as bf16 CUDA tensors; `layers[i]` holds the fused projections."""

    def __init__(self, config: Qwen3Config, device: Optional[torch.device] = None) -> None:
        self.cfg = config
        self.device = device or torch.device("cuda", 0)
        self.embed: Optional[torch.Tensor] = None
        self.final_norm: Optional[torch.Tensor] = None
        self.layers: List[Dict[str, torch.Tensor]] = []
        self.tokenizer = None
        self._cos_sin = self._rope_table(config).to(self.device)

    # -- construction -------------------------------------------------------------------------
    @staticmethod
    def _rope_table(cfg: Qwen3Config) -> torch.Tensor:
        half = cfg.head_dim // 2
        inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, half, dtype=torch.float32) * 2.0 / cfg.head_dim))
        ang = torch.arange(cfg.max_length, dtype=torch.float32)[:, None] * inv_freq[None, :]
        return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous()  # [max_len, 64, 2] fp32

    @classmethod
    def random_init(cls, config: Qwen3Config, seed: int = 0, device: Optional[torch.device] = None,
                    std: float = 0.02) -> "Qwen3Encoder":
        """Seeded random weights at the exact architecture (no checkpoint is reachable offline;
        throughput is value-independent).  Generated on the device, layer by layer."""
        enc = cls(config, device)
        g = torch.Generator(device=enc.device).manual_seed(seed)

        def w(*shape, scale=std):
            return (torch.randn(*shape, generator=g, device=enc.device, dtype=torch.float32) * scale).to(torch.bfloat16)

        def norm_w(n):
            return (1.0 + 0.1 * torch.randn(n, generator=g, device=enc.device, dtype=torch.float32)).to(torch.bfloat16)

        c = config
        enc.embed = w(c.vocab_size, c.hidden_size)
        for _ in range(c.num_layers):
            enc.layers.append({
                "ln1": norm_w(c.hidden_size), "ln2": norm_w(c.hidden_size),
                "qkv": w(c.q_size + 2 * c.kv_size, c.hidden_size), "o": w(c.hidden_size, c.q_size),
                "q_norm": norm_w(c.head_dim), "k_norm": norm_w(c.head_dim),
                "gate_up": w(2 * c.intermediate_size, c.hidden_size), "down": w(c.hidden_size, c.intermediate_size),
            })
        enc.final_norm = norm_w(c.hidden_size)
        return enc

    @classmethod
    def from_state_dict(cls, config: Qwen3Config, sd: Dict[str, torch.Tensor],
                        device: Optional[torch.device] = None, prefix: str = "") -> "Qwen3Encoder":
        """Load weights named like transformers' Qwen3Model (`layers.N.self_attn.q_proj.weight` ...)."""
        enc = cls(config, device)

        def get(name):
            return sd[prefix + name].to(device=enc.device, dtype=torch.bfloat16).contiguous()

        enc.embed = get("embed_tokens.weight")
        for i in range(config.num_layers):
            p = f"layers.{i}."
            enc.layers.append({
                "ln1": get(p + "input_layernorm.weight"), "ln2": get(p + "post_attention_layernorm.weight"),
                "qkv": torch.cat([get(p + "self_attn.q_proj.weight"), get(p + "self_attn.k_proj.weight"),
                                  get(p + "self_attn.v_proj.weight")], dim=0).contiguous(),
                "o": get(p + "self_attn.o_proj.weight"),
                "q_norm": get(p + "self_attn.q_norm.weight"), "k_norm": get(p + "self_attn.k_norm.weight"),
                "gate_up": torch.cat([get(p + "mlp.gate_proj.weight"), get(p + "mlp.up_proj.weight")], dim=0).contiguous(),
                "down": get(p + "mlp.down_proj.weight"),
            })
        enc.final_norm = get("norm.weight")
        return enc

    @classmethod
    def from_pretrained(cls, path: str, device: Optional[torch.device] = None, *, out_dim: Optional[int] = None,
                        max_length: int = 1024, pooling: str = "last", model_id: Optional[str] = None
                        ) -> "Qwen3Encoder":
        """Local directory with config.json, *.safetensors and tokenizer files (nothing is fetched): the gateway's
        startup (RUNBOOK:657-660: AutoTokenizer.from_pretrained(TOKENIZER_PATH, local_files_only=True)) plus the
        checkpoint the ONNX export was made from.  Tensor names are transformers' (`layers.N...` of Qwen3Model, or
        `model.layers.N...` of a *ForCausalLM checkpoint; `lm_head.*` is ignored).  out_dim: EMBED_OUTPUT_DIM
        (default min(1024, hidden)); max_length: EMBED_MAX_LENGTH (RUNBOOK:484)."""
        import json
        from pathlib import Path

        from safetensors.torch import load_file
        root = Path(path)
        hf = json.loads((root / "config.json").read_text())
        rope = hf.get("rope_theta") or (hf.get("rope_parameters") or {}).get("rope_theta", 1_000_000.0)
        cfg = Qwen3Config(hidden_size=hf["hidden_size"], num_layers=hf["num_hidden_layers"],
                          num_heads=hf["num_attention_heads"], num_kv_heads=hf["num_key_value_heads"],
                          head_dim=hf.get("head_dim", 128), intermediate_size=hf["intermediate_size"],
                          vocab_size=hf["vocab_size"], rms_norm_eps=hf.get("rms_norm_eps", 1e-6), rope_theta=rope,
                          max_length=int(max_length), out_dim=int(out_dim or min(1024, hf["hidden_size"])),
                          pooling=pooling, model_id=model_id or hf.get("_name_or_path") or str(root.name))
        if cfg.out_dim > cfg.hidden_size:
            raise ValueError(f"out_dim {cfg.out_dim} exceeds the model's hidden size {cfg.hidden_size}")
        files = sorted(root.glob("*.safetensors"))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {root}")
        sd: Dict[str, torch.Tensor] = {}
        for f in files:
            sd.update(load_file(str(f)))
        prefix = "model." if any(k.startswith("model.") for k in sd) else ""
        enc = cls.from_state_dict(cfg, sd, device, prefix=prefix)
        from transformers import AutoTokenizer
        enc.tokenizer = AutoTokenizer.from_pretrained(str(root), local_files_only=True)
        return enc

    # -- 16 / 32 tokens: the linear layers as weight streams ------------------------------------------------
    def _skinny_weights(self) -> Optional[List[Dict[str, torch.Tensor]]]:
        """The four projections of every layer a second time, in the tile order the weight-streaming kernels read
        (+ 100 % of the layer weights in HBM: 8 GB for the 4B model), built on first use.  None when the model's
        widths are not the ones the kernels are built for or CRAG_ENC_NO_SKINNY is set.  Default: the five-launch
        layer of csrc/crag_encoder_small.hip (tile heights 12 / 10 / 16 / 10 rows for qkv / o / gate|up / down, so
        that the tiles of a projection cover the 256 CUs evenly); CRAG_ENC_SMALL_V1=1 keeps round 3's first version
        (crag_enc_skinny_gemm, 16-row tiles, eight launches per layer) for A/B measurements."""
        c = self.cfg
        off = os.environ.get("CRAG_ENC_NO_SKINNY") is not None or self.__dict__.get("_skinny_failed", False)
        if off != self.__dict__.get("_skinny_off", False):
            self._skinny_off = off
            self.__dict__.pop("_graphs", None)   # graphs captured over the other set of kernels
        if (off or c.hidden_size != 2560 or c.q_size != 4096
                or c.intermediate_size != 9728 or (c.q_size + 2 * c.kv_size) != 6144 or c.head_dim != 128):
            return None
        v1 = os.environ.get("CRAG_ENC_SMALL_V1") is not None
        if self.__dict__.get("_skinny") is None or self.__dict__.get("_skinny_v1") != v1:
            self._skinny = None
            self.__dict__.pop("_graphs", None)   # graphs captured over the other kernels
            try:
                self._build_skinny(v1)
            except torch.OutOfMemoryError:
                # + 100 % of the layer weights does not fit beside what already lives in HBM (an index, its mirror):
                # the library GEMMs answer short queries from now on -- slower, not an error in the middle of a request
                self._skinny = None
                self._skinny_failed = True
                torch.cuda.empty_cache()
                return self._skinny_weights()
            self._skinny_v1 = v1
        return self._skinny

    def _wide_weights(self):
        """All four projections a second time in the fragment order crag_enc_wide_gemm streams (+ 7.3 GB for the 4B model),
        for forwards of exactly 64 or 128 token rows (3 to 8 queries of <= 16 tokens: the gateway's batch sizes,
        RUNBOOK:304,331-334).  Those are the projections where the weight-streaming kernel beats the library's small-M
        GEMM (profiles/r04_wide_gemm.txt, per layer at 128 rows: gate|up + SwiGLU 29.0 vs 37.2 us in ONE launch, down 24.1
        vs 39.3; at 64 rows gate|up 23.5 vs 30.9, down 18.9 vs 21.7); qkv and o stay with the library.  None when the widths are not the 4B
        model's, CRAG_ENC_NO_WIDE is set or the copies do not fit."""
        c = self.cfg
        off = os.environ.get("CRAG_ENC_NO_WIDE") is not None or self.__dict__.get("_wide_failed", False)
        if off != self.__dict__.get("_wide_off", False):
            self._wide_off = off
            self.__dict__.pop("_graphs", None)   # graphs captured over the other set of kernels
        if off or c.hidden_size != 2560 or c.intermediate_size != 9728:
            return None
        if self.__dict__.get("_wide") is None:
            try:
                self._wide = [{"gate_up": ops.wide_gate_up_weight(L["gate_up"]), "down": ops.wide_weight(L["down"]),
                               "o": ops.wide_weight(L["o"]), "qkv": ops.wide_weight(L["qkv"])} for L in self.layers]
            except torch.OutOfMemoryError:
                self._wide, self._wide_failed = None, True
                torch.cuda.empty_cache()
                return self._wide_weights()
        return self._wide

    def warm_up(self) -> None:
        """Build the re-tiled weight copies of the short-query path NOW (at load time, where an allocation failure is a
        start-up event) instead of inside the first short /retrieve request (seconds of re-tiling under the encoder's lock)."""
        self._skinny_weights()
        self._wide_weights()

    def _build_skinny(self, v1: bool) -> None:
        if v1:
            self._skinny = [{"qkv": ops.skinny_weight(L["qkv"]), "o": ops.skinny_weight(L["o"]),
                             "gate_up": ops.skinny_gate_up_weight(L["gate_up"]), "down": ops.skinny_weight(L["down"])}
                            for L in self.layers]
        else:
            self._skinny = [{"qkv": ops.small_weight(L["qkv"], 12), "o": ops.small_weight(L["o"], 10),
                             "gate_up": ops.skinny_gate_up_weight(L["gate_up"]), "down": ops.small_weight(L["down"], 10)}
                            for L in self.layers]

    @torch.no_grad()
    def _forward_small_rows(self, x: torch.Tensor, batch: PackedBatch, skinny) -> torch.Tensor:
        """The 36 layers at 16 or 32 token rows, five launches each (csrc/crag_encoder_small.hip): the residual add and
        the RMSNorm run in the prologue of the projection that consumes them, q/k-norm + RoPE inside the attention
        kernel.  x: [T, hidden] embedding rows (consumed as the first residual stream)."""
        c, dev, bf = self.cfg, self.device, torch.bfloat16
        t = batch.n_tokens
        width = c.q_size + 2 * c.kv_size
        res_a, res_b = x, torch.empty_like(x)
        zeros = self.__dict__.get("_zero_delta")                         # "no delta yet" in front of layer 0: a buffer
        if zeros is None or zeros.shape[0] < t:                          # of zeros nobody writes, not a fill per forward
            zeros = self._zero_delta = torch.zeros(32, c.hidden_size, dtype=bf, device=dev)
        delta = zeros[:t]
        delta_o = torch.empty_like(delta)
        delta_d = torch.empty_like(delta)
        qkv = torch.empty(t, width, dtype=bf, device=dev)
        attn = torch.empty(t, c.q_size, dtype=bf, device=dev)
        act = torch.empty(t, c.intermediate_size, dtype=bf, device=dev)
        scale = 1.0 / math.sqrt(c.head_dim)
        cs_tok = batch.cs_tok if batch.cs_tok is not None else self._cos_sin.index_select(0, batch.positions.long())
        for i, L in enumerate(self.layers):
            W = skinny[i]
            ops.small_gemm(res_a, W["qkv"], qkv, t, width, 12, delta=delta if i == 0 else delta_d, norm_w=L["ln1"],
                           res_out=res_b, eps=c.rms_norm_eps)
            ops.small_attention(qkv, L["q_norm"], L["k_norm"], cs_tok, batch.positions, attn, c.num_heads,
                                c.num_kv_heads, c.rms_norm_eps, scale, by_token=True)
            ops.small_gemm(attn, W["o"], delta_o, t, c.hidden_size, 10)
            ops.small_gemm(res_b, W["gate_up"], act, t, 2 * c.intermediate_size, 16, swiglu=True, delta=delta_o,
                           norm_w=L["ln2"], res_out=res_a, eps=c.rms_norm_eps)
            ops.small_gemm(act, W["down"], delta_d, t, c.hidden_size, 10)
        out = torch.empty(batch.n_seqs, c.out_dim, dtype=torch.float32, device=dev)
        if c.pooling == "last":
            # the pooled rows are DATA (batch.last_tok: in a graph replay the real last token of every padded sequence)
            ops.pool_normalize_rows(res_a, self.final_norm, batch.last_tok, out, c.out_dim, c.rms_norm_eps, delta=delta_d)
        else:
            normed = torch.empty_like(x)
            ops.rmsnorm(delta_d, self.final_norm, normed, c.rms_norm_eps, residual_in=res_a, residual_out=None)
            ops.pool_normalize(normed, None, batch.cu, out, c.out_dim, 1, c.rms_norm_eps)
        return out

    # -- forward --------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_packed(self, ids: torch.Tensor, batch: PackedBatch) -> torch.Tensor:
        """ids: int32 [T] on the device.  Returns unit-norm embeddings [B, out_dim] fp32."""
        c = self.cfg
        if c.head_dim != 128:
            raise ValueError("the HIP attention / rope kernels are specialised for head_dim 128")
        t, dev = batch.n_tokens, self.device
        bf = torch.bfloat16
        x = torch.empty(t, c.hidden_size, dtype=bf, device=dev)
        ops.embed_gather(ids, self.embed, x)
        # 16 or 32 tokens (one short query): the projections stream their weights (crag_encoder_small.hip)
        # 32 rows (ONE query of 17..32 tokens, or two of <= 16): gate|up and down through the wide weight-streaming kernels
        # at 32 rows, qkv / o through the library, one attention launch -- 2.95-3.04 ms against 3.58 through the
        # five-launch layer's 32-row kernels and 3.34-3.42 padded to 64 rows (profiles/r04_pad32_encode.txt);
        # CRAG_ENC_NO_WIDE_32=1 (or no room for the wide copies) keeps the five-launch layer
        wide32 = (t == 32 and os.environ.get("CRAG_ENC_NO_WIDE_32") is None and os.environ.get("CRAG_ENC_SMALL_V1") is None
                  and self._wide_weights() is not None)
        skinny = self._skinny_weights() if t in (16, 32) and not wide32 else None
        if skinny is not None and not self._skinny_v1:
            return self._forward_small_rows(x, batch, skinny)
        resid = torch.empty_like(x)
        normed = torch.empty_like(x)
        width = c.q_size + 2 * c.kv_size
        qkv_buf = torch.zeros(t + 32, width, dtype=bf, device=dev)  # attention reads up to 31 rows past T
        qkv = qkv_buf[:t]
        vt = torch.empty(c.num_kv_heads, c.head_dim, batch.t_pad, dtype=bf, device=dev)
        attn = torch.empty(t, c.q_size, dtype=bf, device=dev)
        act = torch.empty(t, c.intermediate_size, dtype=bf, device=dev)
        scale = 1.0 / math.sqrt(c.head_dim)
        last_only = c.pooling == "last" and os.environ.get("CRAG_ENC_FULL_LAST_LAYER") is None
        full_last_wide = False   # (set below: the wide path keeps ALL rows in the last layer)
        wide = self._wide_weights() if t in (64, 96, 128) or wide32 else None
        short_seqs = (0 < batch.max_len <= 32 and batch.n_seqs <= 65535 and skinny is None
                      and os.environ.get("CRAG_ENC_NO_SHORT_ATTN") is None)
        cs_tok = None
        if short_seqs:   # once per forward; in the graph path once per captured shape
            cs_tok = batch.cs_tok if batch.cs_tok is not None else self._cos_sin.index_select(0, batch.positions.long())
        delta: Optional[torch.Tensor] = None  # output of the previous sub-block, added into the residual
        down_parts, down_split = None, (8 if t >= 96 else 4)
        if wide is not None and last_only and os.environ.get("CRAG_ENC_MIX_LAST_ONLY") is None:
            # 32..128 rows through the weight-streaming kernels: the last layer's o / gate|up / down cost the same for
            # 128 rows as for the 1..8 pooled ones (the weights are the stream), and the library's GEMMs at 1..8 rows
            # stream them slower -- all rows stay in, the pooled rows are picked at the end
            last_only, full_last_wide = False, True
        fuse_reduce = os.environ.get("CRAG_ENC_NO_FUSED_REDUCE") is None
        for i, L in enumerate(self.layers):
            if i == 0:
                ops.rmsnorm(x, L["ln1"], normed, c.rms_norm_eps, residual_in=None, residual_out=None)
                resid.copy_(x)
            elif down_parts is not None:   # the previous layer's down left its split-K partial tiles: summed by the norm
                ops.rmsnorm_partials(down_parts, down_split, t, L["ln1"], normed, c.rms_norm_eps, residual_in=resid,
                                     residual_out=resid)
                down_parts = None
            else:
                ops.rmsnorm(delta, L["ln1"], normed, c.rms_norm_eps, residual_in=resid, residual_out=resid)
            # hipBLASLt picks a ~13 % faster kernel for this shape (K 2560, N 6144) at M = 32768 than at 65536
            # (scripts/probes/gemm_layouts.py), so large batches run the projection in row chunks
            qkv_parts = None
            if wide is not None and fuse_reduce and short_seqs and os.environ.get("CRAG_ENC_NO_WIDE_QKV") is None:
                # qkv with K split 2 ways through the wide kernel; the attention kernel sums the two partial tiles of
                # every head vector while it loads them
                qkv_parts = ops.wide_gemm_rows(normed, wide[i]["qkv"], width, 2)
            elif skinny is not None:
                ops.skinny_gemm(normed, skinny[i]["qkv"], qkv, t, width)
            for lo in range(0, t if skinny is None and qkv_parts is None else 0, QKV_ROW_CHUNK):
                hi = min(t, lo + QKV_ROW_CHUNK)
                torch.matmul(normed[lo:hi], L["qkv"].t(), out=qkv[lo:hi])
            if qkv_parts is not None:
                ops.small_attention_seqs_parts(qkv_parts, 2, t, L["q_norm"], L["k_norm"], cs_tok, batch.positions, batch.cu,
                                               batch.n_seqs, batch.max_len, attn, c.num_heads, c.num_kv_heads,
                                               c.rms_norm_eps, scale, by_token=True)
            elif short_seqs:
                # every sequence <= 32 tokens (a batch of short queries): q/k-norm + RoPE + attention in ONE launch, a
                # workgroup per (q head, sequence), instead of the rope / V-transpose launch + the flash kernel
                ops.small_attention_seqs(qkv, L["q_norm"], L["k_norm"], cs_tok, batch.positions, batch.cu, batch.n_seqs,
                                         batch.max_len, attn, c.num_heads, c.num_kv_heads, c.rms_norm_eps, scale,
                                         by_token=True)
            else:
                ops.qk_rope_vt(qkv_buf, L["q_norm"], L["k_norm"], self._cos_sin, batch.positions,
                               c.num_heads, c.num_kv_heads, c.rms_norm_eps, vt, batch.tok_of_pad)
                ops.attention(qkv_buf, vt, attn, batch.cu, batch.cu_pad, batch.blk_seq, batch.blk_q0,
                              c.num_heads, c.num_kv_heads, scale)
            if last_only and i == len(self.layers) - 1:
                # Last layer, last-token pooling: behind the attention only the B pooled rows matter (no later
                # layer reads the other tokens' hidden states), so the output projection, the second norm and the
                # whole MLP run on B rows instead of T: one layer's worth of o / gate|up / down GEMMs less.
                attn_l = attn.index_select(0, batch.last_tok)
                resid_l = resid.index_select(0, batch.last_tok)
                delta_l = F.linear(attn_l, L["o"])
                normed_l = torch.empty_like(resid_l)
                ops.rmsnorm(delta_l, L["ln2"], normed_l, c.rms_norm_eps, residual_in=resid_l, residual_out=resid_l)
                act_l = torch.empty(batch.n_seqs, c.intermediate_size, dtype=bf, device=dev)
                ops.swiglu(F.linear(normed_l, L["gate_up"]), act_l)
                delta_l = F.linear(act_l, L["down"])
                out = torch.empty(batch.n_seqs, c.out_dim, dtype=torch.float32, device=dev)
                ops.pool_normalize(resid_l, self.final_norm, batch.cu_one, out, c.out_dim, 0, c.rms_norm_eps, delta=delta_l)
                return out
            if skinny is not None:
                delta = ops.skinny_gemm(attn, skinny[i]["o"], torch.empty(t, c.hidden_size, dtype=bf, device=dev), t,
                                        c.hidden_size)
                ops.rmsnorm(delta, L["ln2"], normed, c.rms_norm_eps, residual_in=resid, residual_out=resid)
                ops.skinny_gemm(normed, skinny[i]["gate_up"], act, t, 2 * c.intermediate_size, swiglu=True)
                delta = ops.skinny_gemm(act, skinny[i]["down"], torch.empty(t, c.hidden_size, dtype=bf, device=dev), t,
                                        c.hidden_size)
                continue
            if wide is not None and fuse_reduce:
                # o with K split 4 ways through the wide kernel, its token-major partial tiles summed by the norm that
                # consumes them: the library's GEMM + a norm launch (at 128 rows 18.8 + 5 us) become 13 + 5
                o_parts = ops.wide_gemm_rows(attn, wide[i]["o"], c.hidden_size, 4)
                ops.rmsnorm_partials(o_parts, 4, t, L["ln2"], normed, c.rms_norm_eps, residual_in=resid, residual_out=resid)
            else:
                delta = F.linear(attn, L["o"])
                ops.rmsnorm(delta, L["ln2"], normed, c.rms_norm_eps, residual_in=resid, residual_out=resid)
            if wide is not None:   # 64 / 128 rows: gate|up + SwiGLU as ONE weight-streaming launch (crag_encoder_wide.hip)
                ops.wide_gemm(normed, wide[i]["gate_up"], act, t, 2 * c.intermediate_size, 1, swiglu=True)
            else:
                gate_up = F.linear(normed, L["gate_up"])
                ops.swiglu(gate_up, act)
            if wide is not None and i + 1 < len(self.layers) and fuse_reduce:
                # ... and down with K split 8 ways (128 rows) / 4 ways (64 and 32 rows, 64-row tiles); its partial tiles
                # stay in fp32, token-major, for the next layer's norm to sum (no reduce launch)
                down_parts, delta = ops.wide_gemm_rows(act, wide[i]["down"], c.hidden_size, down_split), None
            elif wide is not None:
                delta = ops.wide_gemm(act, wide[i]["down"], torch.empty(t, c.hidden_size, dtype=bf, device=dev), t,
                                      c.hidden_size, down_split)
            else:
                delta = F.linear(act, L["down"])
        out = torch.empty(batch.n_seqs, c.out_dim, dtype=torch.float32, device=dev)
        if c.pooling == "last" and full_last_wide:
            # the pooled rows are DATA (batch.last_tok: in a graph replay the real last token of every padded sequence)
            ops.pool_normalize_rows(resid, self.final_norm, batch.last_tok, out, c.out_dim, c.rms_norm_eps, delta=delta)
        elif c.pooling == "last":
            # residual + last delta and the final norm for the pooled rows only, inside the pool kernel
            ops.pool_normalize(resid, self.final_norm, batch.cu, out, c.out_dim, 0, c.rms_norm_eps, delta=delta)
        elif c.pooling == "mean":
            ops.rmsnorm(delta, self.final_norm, normed, c.rms_norm_eps, residual_in=resid, residual_out=None)
            ops.pool_normalize(normed, None, batch.cu, out, c.out_dim, 1, c.rms_norm_eps)
        else:
            raise ValueError(f"unknown pooling {c.pooling!r}")
        return out

    # -- small batches: one hipGraph replay per forward ---------------------------------------------------
    # A /retrieve request embeds ONE query (/root/reference/app/retrieve.py:427).  At 16 tokens the eager forward is
    # ~470 kernel launches from Python (36 layers x 13) = 5 ms for 25 us of weight streaming per layer; captured
    # once per shape it is one graph launch.  Shape = (sequences, bucket length): every sequence is padded BEHIND
    # its last real token up to the bucket (causal attention: a real token never sees a later pad) and pooled at its
    # real last token, whose position is DATA (a device tensor), not shape.
    SMALL_BUCKETS = (16, 32, 64, 128, 256, 512, 1024)
    SMALL_MAX_TOKENS = 2048      # padded tokens per forward that still take this path
    SMALL_FREE_TOKENS = 128      # up to here the forward streams weights: padded tokens cost nothing
    SMALL_MAX_WASTE = 1.25       # beyond it: at most this many padded tokens per real token
    SMALL_MAX_GRAPHS = 12

    def _small_bucket(self, lens: Sequence[int]) -> Optional[int]:
        if (self.cfg.pooling != "last" or not lens or os.environ.get("CRAG_ENC_NO_GRAPH") is not None
                or os.environ.get("CRAG_ENC_FULL_LAST_LAYER") is not None):
            return None
        longest, real = max(lens), sum(lens)
        for b in self.SMALL_BUCKETS:
            if longest <= b:
                padded = b * len(lens)
                ok = padded <= self.SMALL_FREE_TOKENS or (padded <= self.SMALL_MAX_TOKENS and
                                                          padded <= self.SMALL_MAX_WASTE * real)
                return b if ok else None
        return None

    def _small_graph(self, n_seqs: int, bucket: int):
        cache = self.__dict__.setdefault("_graphs", {})
        key = (n_seqs, bucket)
        hit = cache.get(key)
        if hit is not None:
            hit["used"] = self.__dict__["_graph_clock"] = self.__dict__.get("_graph_clock", 0) + 1
            return hit
        if len(cache) >= self.SMALL_MAX_GRAPHS:   # drop the least recently used graph (and its private pool)
            torch.cuda.synchronize(self.device)   # ... once no replay of it can still be running on any stream
            del cache[min(cache, key=lambda k: cache[k]["used"])]
        batch = PackedBatch.build([bucket] * n_seqs, self.device)
        # the two inputs of a replay -- token ids (int32 [T]) and the pooled rows (int64 [B]) -- live in ONE device
        # buffer with ONE pinned twin: one upload per forward (two cost a one-query forward a second copy and a second
        # 18-us host gap in front of the graph: profiles/r04_small_layer_kernel_trace.txt)
        t_rows = n_seqs * bucket
        off_last = (4 * t_rows + 7) // 8 * 8
        d_in = torch.zeros(off_last + 8 * n_seqs, dtype=torch.uint8, device=self.device)
        ids = d_in[:4 * t_rows].view(torch.int32)
        batch.last_tok = d_in[off_last:].view(torch.int64)
        batch.last_tok.copy_(torch.arange(1, n_seqs + 1, dtype=torch.int64, device=self.device) * bucket - 1)
        batch.cs_tok = self._cos_sin.index_select(0, batch.positions.long())   # outside the graph: a constant of the shape
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):          # warm-up outside the capture (library handles, autotuning)
            self.forward_packed(ids, batch)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.forward_packed(ids, batch)
        # pinned staging for the two small uploads of a replay + the event of the last upload (an async copy from a
        # pageable temporary is only safe while the runtime happens to stage it before returning)
        # TWO pinned twins used in turn, each with the event of the replay that last read it, recorded BEHIND the
        # replay: no marker packet between the upload and the graph, and the host fills the other buffer while a
        # forward is still running
        def pinned():
            h = torch.zeros(off_last + 8 * n_seqs, dtype=torch.uint8).pin_memory()
            return {"h_in": h, "h_ids": h[:4 * t_rows].view(torch.int32), "h_last": h[off_last:].view(torch.int64),
                    "read": torch.cuda.Event()}
        entry = {"graph": graph, "ids": ids, "last_tok": batch.last_tok, "out": out, "batch": batch,
                 "d_in": d_in, "pinned": [pinned(), pinned()], "turn": 0,
                 "used": self.__dict__.get("_graph_clock", 0)}
        cache[key] = entry
        return entry

    @torch.no_grad()
    def _forward_small(self, token_lists: Sequence[Sequence[int]], lens: Sequence[int], bucket: int) -> torch.Tensor:
        n = len(lens)
        # 33..127 padded rows that are not a multiple of 32: phantom sequences round the batch up to 64 / 96 / 128 rows,
        # the heights the weight-streaming kernels are built for (3 queries of 16 tokens: 3.66 ms through the library's
        # GEMMs at 48 rows, 3.16 ms as 64 rows; 5 queries 3.94 -> as 96 rows; 7 queries 4.48 -> 3.95 ms) -- their outputs
        # are dropped below.  (32 / 64 / 96 / 128 rows go as they are, forward_packed.)
        n_real = n
        if bucket <= 64 and (32 < n * bucket < 128) and n * bucket not in (64, 96) and self._wide_weights() is not None:
            target = 64 if n * bucket < 64 else (96 if n * bucket < 96 else 128)
            if target % bucket:          # (bucket 64: 96 rows do not divide)
                target = 128
            if target % bucket == 0:
                n = target // bucket
                token_lists = list(token_lists) + [[0]] * (n - n_real)
                lens = list(lens) + [1] * (n - n_real)
        if n * bucket in (32, 64, 96, 128):
            self._wide_weights()        # (a flip of CRAG_ENC_NO_WIDE drops the graphs captured over the other kernels)
        if n * bucket in (16, 32):
            self._skinny_weights()      # a flip of CRAG_ENC_SMALL_V1 / CRAG_ENC_NO_SKINNY drops the graphs captured over the other kernels
        g = self._small_graph(n, bucket)
        pin = g["pinned"][g["turn"]]
        g["turn"] ^= 1
        pin["read"].synchronize()              # the replay before last has copied this buffer out
        host = pin["h_ids"].numpy().reshape(n, bucket)
        host.fill(0)
        for i, (tl, m) in enumerate(zip(token_lists, lens)):
            host[i, :m] = np.asarray(tl[:m], dtype=np.int32)
        pin["h_last"].numpy()[:] = np.arange(n, dtype=np.int64) * bucket + (np.asarray(lens, dtype=np.int64) - 1)
        g["d_in"].copy_(pin["h_in"], non_blocking=True)
        g["graph"].replay()
        pin["read"].record()
        return g["out"][:n_real].clone()

    @torch.no_grad()
    def embed_token_lists(self, token_lists: Sequence[Sequence[int]]) -> torch.Tensor:
        lens = [min(len(tl), self.cfg.max_length) for tl in token_lists]
        if any(n <= 0 for n in lens):
            raise ValueError("every sequence needs at least one token")
        bucket = self._small_bucket(lens)
        if bucket is not None:
            return self._forward_small(token_lists, lens, bucket)
        flat = np.concatenate([np.asarray(tl[:n], dtype=np.int32) for tl, n in zip(token_lists, lens)])
        batch = PackedBatch.build(lens, self.device)
        ids = torch.from_numpy(flat).to(self.device)
        return self.forward_packed(ids, batch)

    # -- the Encoder protocol of cadence_rag_amd.embeddings ------------------------------------------
    def tokenize(self, texts: Sequence[str]) -> List[List[int]]:
        """The gateway's tokenizer call (RUNBOOK:689-699: truncation=True, max_length=EMBED_MAX_LENGTH) without its
        padding: sequences are packed, not padded."""
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer loaded (Qwen3Encoder.from_pretrained, or set .tokenizer)")
        return self.tokenizer(list(texts), truncation=True, max_length=self.cfg.max_length, padding=False)["input_ids"]

    def encode_device(self, texts: Sequence[str]) -> Tuple[torch.Tensor, str]:
        """texts -> unit-norm embeddings [n, out_dim] fp32 ON THE DEVICE (embeddings.embed_texts_device)."""
        return self.embed_token_lists(self.tokenize(texts)), self.cfg.model_id

    def encode(self, texts: Sequence[str]) -> Tuple[List[List[float]], str]:
        vecs, model = self.encode_device(texts)
        return vecs.cpu().tolist(), model


class ByteTokenizer:
    """Deterministic stand-in tokenizer for tests and synthetic benchmarks ONLY (utf-8 bytes + an
    end marker).  Real deployments load the model's own tokenizer with from_pretrained."""

    def __init__(self, eos_id: int = 256) -> None:
        self.eos_id = eos_id

    def __call__(self, texts, truncation=True, max_length=1024, padding=False):
        out = []
        for t in texts:
            ids = list(t.encode("utf-8"))[: max_length - 1] + [self.eos_id]
            out.append(ids)
        return {"input_ids": out}
