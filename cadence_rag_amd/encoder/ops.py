"""torch-tensor wrappers over the crag_enc_* C ABI (include/crag_encoder.h).  Tensors must be
contiguous CUDA tensors; bf16 tensors are passed as raw storage.  No CPU fallback."""
from __future__ import annotations

import ctypes

import torch

from .. import _native


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous CUDA tensor of dtype {dtype}")


def embed_gather(ids: torch.Tensor, table: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _req(ids, torch.int32, "ids"); _req(table, torch.bfloat16, "table"); _req(out, torch.bfloat16, "out")
    _native.check(_native.load().crag_enc_embed_gather(_p(ids), _p(table), _p(out), ids.numel(), table.shape[1],
                                                       table.shape[0], _stream()), "crag_enc_embed_gather")
    return out


def rmsnorm(x, weight, out, eps: float, residual_in=None, residual_out=None):
    _req(x, torch.bfloat16, "x"); _req(weight, torch.bfloat16, "weight"); _req(out, torch.bfloat16, "out")
    rows, hidden = x.shape
    _native.check(_native.load().crag_enc_rmsnorm(_p(x), _p(residual_in), _p(weight), _p(out), _p(residual_out),
                                                  rows, hidden, float(eps), _stream()), "crag_enc_rmsnorm")
    return out


def qk_norm_rope(qkv, q_w, k_w, cos_sin, positions, hq: int, hkv: int, eps: float):
    _req(qkv, torch.bfloat16, "qkv"); _req(cos_sin, torch.float32, "cos_sin"); _req(positions, torch.int32, "positions")
    _native.check(_native.load().crag_enc_qk_norm_rope(_p(qkv), _p(q_w), _p(k_w), _p(cos_sin), _p(positions),
                                                       positions.numel(), hq, hkv, float(eps), _stream()),
                  "crag_enc_qk_norm_rope")
    return qkv


def v_transpose(qkv, vt, tok_of_pad, hq: int, hkv: int):
    _req(qkv, torch.bfloat16, "qkv"); _req(vt, torch.bfloat16, "vt"); _req(tok_of_pad, torch.int32, "tok_of_pad")
    _native.check(_native.load().crag_enc_v_transpose(_p(qkv), _p(vt), _p(tok_of_pad), tok_of_pad.numel(), hq, hkv,
                                                      _stream()), "crag_enc_v_transpose")
    return vt


def attention(qkv, vt, out, cu, cu_pad, blk_seq, blk_q0, hq: int, hkv: int, scale: float):
    _req(qkv, torch.bfloat16, "qkv"); _req(vt, torch.bfloat16, "vt"); _req(out, torch.bfloat16, "out")
    _native.check(_native.load().crag_enc_attention(_p(qkv), _p(vt), _p(out), _p(cu), _p(cu_pad), _p(blk_seq),
                                                    _p(blk_q0), blk_seq.numel(), vt.shape[-1], hq, hkv, float(scale),
                                                    _stream()), "crag_enc_attention")
    return out


def swiglu(gate_up, out):
    _req(gate_up, torch.bfloat16, "gate_up"); _req(out, torch.bfloat16, "out")
    rows, two_i = gate_up.shape
    _native.check(_native.load().crag_enc_swiglu(_p(gate_up), _p(out), rows, two_i // 2, _stream()), "crag_enc_swiglu")
    return out


def pool_normalize(hidden_states, final_norm_w, cu, out, out_dim: int, mode: int, eps: float, delta=None):
    _req(hidden_states, torch.bfloat16, "hidden_states"); _req(out, torch.float32, "out")
    if delta is not None:
        _req(delta, torch.bfloat16, "delta")
        if delta.shape != hidden_states.shape:
            raise ValueError("delta must have the shape of hidden_states")
    _native.check(_native.load().crag_enc_pool_normalize_add(_p(hidden_states), _p(delta), _p(final_norm_w), _p(cu),
                                                             _p(out), cu.numel() - 1, hidden_states.shape[1], out_dim,
                                                             mode, float(eps), _stream()), "crag_enc_pool_normalize")
    return out


def pool_normalize_rows(hidden_states, final_norm_w, rows, out, out_dim: int, eps: float, delta=None):
    """Last-token pooling + final RMSNorm + slice + L2 normalise of the rows `rows` (int64 [B], device) of hidden_states
    (+ delta): crag_enc_pool_normalize_rows -- the pooled rows are data, nothing is gathered first."""
    _req(hidden_states, torch.bfloat16, "hidden_states"); _req(out, torch.float32, "out"); _req(rows, torch.int64, "rows")
    if delta is not None:
        _req(delta, torch.bfloat16, "delta")
        if delta.shape != hidden_states.shape:
            raise ValueError("delta must have the shape of hidden_states")
    _native.check(_native.load().crag_enc_pool_normalize_rows(_p(hidden_states), _p(delta), _p(final_norm_w), _p(rows),
                                                              _p(out), rows.numel(), hidden_states.shape[1], out_dim,
                                                              float(eps), _stream()), "crag_enc_pool_normalize_rows")
    return out


def skinny_weight(weight: torch.Tensor) -> torch.Tensor:
    """[n, k] bf16 (torch Linear layout) -> the MFMA A-fragment order crag_enc_skinny_gemm streams
    ([n/16][k/32][lane = 16 (kk/8) + row][8]); n % 16 == 0, k % 32 == 0."""
    n, k = weight.shape
    return weight.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def skinny_gate_up_weight(gate_up: torch.Tensor) -> torch.Tensor:
    """The fused [2I, k] gate|up weight with rows interleaved per 8 features (tile t = gate rows 8t..8t+7, then up rows
    8t..8t+7), in fragment order: the operand of skinny_gemm(..., swiglu=True)."""
    two_i, k = gate_up.shape
    inter = two_i // 2
    g = gate_up[:inter].view(inter // 8, 8, k)
    u = gate_up[inter:].view(inter // 8, 8, k)
    return skinny_weight(torch.cat([g, u], dim=1).reshape(two_i, k))


def skinny_gemm(x: torch.Tensor, wsw: torch.Tensor, out: torch.Tensor, m_rows: int, n: int, swiglu: bool = False):
    """out[m_rows, n (or n/2)] = x[m_pad, k] @ W^T for m_pad = 16 or 32 token rows (crag_enc_skinny_gemm)."""
    _req(x, torch.bfloat16, "x"); _req(wsw, torch.bfloat16, "wsw"); _req(out, torch.bfloat16, "out")
    m_pad, k = x.shape
    _native.check(_native.load().crag_enc_skinny_gemm(_p(x), _p(wsw), _p(out), int(m_rows), int(m_pad), int(n), int(k),
                                                      1 if swiglu else 0, _stream()), "crag_enc_skinny_gemm")
    return out


def qk_rope_vt(qkv, q_w, k_w, cos_sin, positions, hq: int, hkv: int, eps: float, vt, tok_of_pad):
    """qk_norm_rope (in place on q|k) and v_transpose (v -> vt) in one launch."""
    _req(qkv, torch.bfloat16, "qkv"); _req(cos_sin, torch.float32, "cos_sin"); _req(positions, torch.int32, "positions")
    _req(vt, torch.bfloat16, "vt"); _req(tok_of_pad, torch.int32, "tok_of_pad")
    _native.check(_native.load().crag_enc_qk_rope_vt(_p(qkv), _p(q_w), _p(k_w), _p(cos_sin), _p(positions),
                                                     positions.numel(), hq, hkv, float(eps), _p(vt), _p(tok_of_pad),
                                                     tok_of_pad.numel(), _stream()), "crag_enc_qk_rope_vt")
    return qkv


# -- the layer at 16 / 32 token rows as five launches (csrc/crag_encoder_small.hip) -------------------------------
def small_weight(weight: torch.Tensor, rows: int) -> torch.Tensor:
    """[n, k] bf16 (torch Linear layout) -> the tile order crag_enc_small_gemm streams:
    [n / rows][k / 32][4][rows][8]; n % rows == 0, k % 32 == 0.  rows = 16 is skinny_weight's order."""
    n, k = weight.shape
    if n % rows or k % 32:
        raise ValueError(f"weight [{n}, {k}] does not split into {rows}-row tiles of 32-element k-steps")
    return weight.view(n // rows, rows, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def small_gemm(x: torch.Tensor, wsw: torch.Tensor, out: torch.Tensor, m_rows: int, n: int, rows: int, *,
               swiglu: bool = False, delta=None, norm_w=None, res_out=None, eps: float = 1e-6):
    """out[m_rows, n (or n/2)] = X @ W^T for 16 or 32 rows; with norm_w: X = RMSNorm(x + delta) * norm_w computed in
    the kernel's prologue, res_out <- x + delta (crag_enc_small_gemm)."""
    _req(x, torch.bfloat16, "x"); _req(wsw, torch.bfloat16, "wsw"); _req(out, torch.bfloat16, "out")
    for name, t in (("delta", delta), ("norm_w", norm_w), ("res_out", res_out)):
        if t is not None:
            _req(t, torch.bfloat16, name)
    m_pad, k = x.shape
    _native.check(_native.load().crag_enc_small_gemm(_p(x), _p(delta), _p(norm_w), _p(res_out), _p(wsw), _p(out),
                                                     int(m_rows), int(m_pad), int(n), int(k), int(rows),
                                                     1 if swiglu else 0, float(eps), _stream()), "crag_enc_small_gemm")
    return out


def small_attention(qkv, q_w, k_w, cos_sin, positions, out, hq: int, hkv: int, eps: float, scale: float,
                    by_token: bool = False):
    """q/k-norm + RoPE + causal attention of <= 32 packed token rows (crag_enc_small_attention); qkv is not modified.
    by_token: cos_sin holds the table rows of the tokens' positions, [T, 64, 2]."""
    _req(qkv, torch.bfloat16, "qkv"); _req(cos_sin, torch.float32, "cos_sin"); _req(positions, torch.int32, "positions")
    _req(out, torch.bfloat16, "out"); _req(q_w, torch.bfloat16, "q_w"); _req(k_w, torch.bfloat16, "k_w")
    _native.check(_native.load().crag_enc_small_attention(_p(qkv), _p(q_w), _p(k_w), _p(cos_sin), 1 if by_token else 0,
                                                          _p(positions), _p(out),
                                                          positions.numel(), hq, hkv, float(eps), float(scale),
                                                          _stream()), "crag_enc_small_attention")
    return out


def small_attention_seqs(qkv, q_w, k_w, cos_sin, positions, cu, n_seqs: int, max_len: int, out, hq: int, hkv: int,
                         eps: float, scale: float, by_token: bool = False):
    """q/k-norm + RoPE + causal attention of a packed batch of short sequences (every one <= max_len <= 32 tokens) in
    one launch, one workgroup per (q head, sequence) (crag_enc_small_attention_seqs); qkv is not modified."""
    _req(qkv, torch.bfloat16, "qkv"); _req(cos_sin, torch.float32, "cos_sin"); _req(positions, torch.int32, "positions")
    _req(out, torch.bfloat16, "out"); _req(q_w, torch.bfloat16, "q_w"); _req(k_w, torch.bfloat16, "k_w")
    _req(cu, torch.int32, "cu")
    _native.check(_native.load().crag_enc_small_attention_seqs(_p(qkv), _p(q_w), _p(k_w), _p(cos_sin),
                                                               1 if by_token else 0, _p(positions), _p(cu), int(n_seqs),
                                                               int(max_len), _p(out), hq, hkv, float(eps), float(scale),
                                                               _stream()), "crag_enc_small_attention_seqs")
    return out


def small_attention_seqs_parts(parts, splitk: int, m_pad: int, q_w, k_w, cos_sin, positions, cu, n_seqs: int, max_len: int,
                               out, hq: int, hkv: int, eps: float, scale: float, by_token: bool = False):
    """small_attention_seqs over the qkv projection's split-K partial tiles (wide_gemm_rows): the head vectors are summed
    over the splits and rounded to bf16 while they are loaded (crag_enc_small_attention_seqs_parts)."""
    _req(parts, torch.float32, "parts"); _req(cos_sin, torch.float32, "cos_sin"); _req(positions, torch.int32, "positions")
    _req(out, torch.bfloat16, "out"); _req(q_w, torch.bfloat16, "q_w"); _req(k_w, torch.bfloat16, "k_w")
    _req(cu, torch.int32, "cu")
    _native.check(_native.load().crag_enc_small_attention_seqs_parts(_p(parts), int(splitk), int(m_pad), _p(q_w), _p(k_w),
                                                                     _p(cos_sin), 1 if by_token else 0, _p(positions),
                                                                     _p(cu), int(n_seqs), int(max_len), _p(out), hq, hkv,
                                                                     float(eps), float(scale), _stream()),
                  "crag_enc_small_attention_seqs_parts")
    return out


# -- the linear layers at 64 / 128 token rows (csrc/crag_encoder_wide.hip) ------------------------------------------
def wide_weight(weight: torch.Tensor) -> torch.Tensor:
    """[n, k] bf16 (torch Linear layout) -> the A-fragment order of v_mfma_f32_32x32x16_bf16 that crag_enc_wide_gemm
    streams: [n/32][k/16][lane = 32 (kk/8) + row][8]; n % 128 == 0, k % 128 == 0."""
    n, k = weight.shape
    if n % 128 or k % 128:
        raise ValueError(f"weight [{n}, {k}] does not split into 128-row tiles of 128-element chunks")
    return weight.view(n // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def wide_gate_up_weight(gate_up: torch.Tensor) -> torch.Tensor:
    """The fused [2I, k] gate|up weight with every 32 rows = 16 gate rows then the 16 up rows of the same features, in
    fragment order: the operand of wide_gemm(..., swiglu=True)."""
    two_i, k = gate_up.shape
    inter = two_i // 2
    g = gate_up[:inter].view(inter // 16, 16, k)
    u = gate_up[inter:].view(inter // 16, 16, k)
    return wide_weight(torch.cat([g, u], dim=1).reshape(two_i, k))


_wide_scratch: dict = {}


def wide_gemm(x: torch.Tensor, ww: torch.Tensor, out: torch.Tensor, m_rows: int, n: int, splitk: int,
              swiglu: bool = False, scratch: "torch.Tensor | None" = None):
    """out[m_rows, n (or n/2)] = x[m_pad, k] @ W^T for m_pad = 32, 64, 96 or 128 token rows: crag_enc_wide_gemm (split-K partial
    tiles, fp32) + crag_enc_wide_reduce (sum of the splits, one rounding to bf16, optional SwiGLU).  scratch: fp32
    buffer of >= splitk * n * m_pad elements (one per device is kept otherwise: calls on one stream reuse it in order)."""
    _req(x, torch.bfloat16, "x"); _req(ww, torch.bfloat16, "ww"); _req(out, torch.bfloat16, "out")
    m_pad, k = x.shape
    if int(splitk) == 1:   # one launch: the GEMM kernel rounds and writes the output itself
        _native.check(_native.load().crag_enc_wide_gemm_direct(_p(x), _p(ww), _p(out), int(m_rows), int(m_pad), int(n), int(k),
                                                               1 if swiglu else 0, _stream()), "crag_enc_wide_gemm_direct")
        return out
    need = int(splitk) * int(n) * int(m_pad)
    if scratch is None:
        scratch = _wide_scratch.get(x.device)
        if scratch is None or scratch.numel() < need:
            scratch = _wide_scratch[x.device] = torch.empty(max(need, 4 * 19456 * 128), dtype=torch.float32, device=x.device)
    elif scratch.dtype != torch.float32 or scratch.numel() < need:
        raise ValueError("scratch must be a float32 tensor of at least splitk * n * m_pad elements")
    lib = _native.load()
    _native.check(lib.crag_enc_wide_gemm(_p(x), _p(ww), _p(scratch), int(m_pad), int(n), int(k), int(splitk), _stream()),
                  "crag_enc_wide_gemm")
    _native.check(lib.crag_enc_wide_reduce(_p(scratch), _p(out), int(m_rows), int(m_pad), int(n), int(splitk),
                                           1 if swiglu else 0, _stream()), "crag_enc_wide_reduce")
    return out


def wide_gemm_rows(x: torch.Tensor, ww: torch.Tensor, n: int, splitk: int, scratch: "torch.Tensor | None" = None):
    """The split-K half of wide_gemm alone, its fp32 partial tiles TOKEN-MAJOR ([splitk, m_pad, n]) for
    rmsnorm_partials to consume (crag_enc_wide_gemm_rows).  Returns the scratch tensor that holds them: the per-device
    buffer unless one is passed -- valid until the next split-K wide_gemm / wide_gemm_rows on that device."""
    _req(x, torch.bfloat16, "x"); _req(ww, torch.bfloat16, "ww")
    m_pad, k = x.shape
    need = int(splitk) * int(n) * int(m_pad)
    if scratch is None:
        scratch = _wide_scratch.get(x.device)
        if scratch is None or scratch.numel() < need:
            scratch = _wide_scratch[x.device] = torch.empty(max(need, 4 * 19456 * 128), dtype=torch.float32, device=x.device)
    elif scratch.dtype != torch.float32 or scratch.numel() < need:
        raise ValueError("scratch must be a float32 tensor of at least splitk * n * m_pad elements")
    _native.check(_native.load().crag_enc_wide_gemm_rows(_p(x), _p(ww), _p(scratch), int(m_pad), int(n), int(k), int(splitk),
                                                         _stream()), "crag_enc_wide_gemm_rows")
    return scratch


def rmsnorm_partials(partial_rows: torch.Tensor, splitk: int, m_pad: int, weight: torch.Tensor, out: torch.Tensor,
                     eps: float, residual_in: torch.Tensor, residual_out: "torch.Tensor | None"):
    """out = RMSNorm(residual_in + bf16(sum of the splitk token-major partial tiles)) * weight; residual_out <- the sum
    (crag_enc_rmsnorm_partials: wide_reduce + rmsnorm in one launch, the same roundings)."""
    _req(partial_rows, torch.float32, "partial_rows"); _req(weight, torch.bfloat16, "weight"); _req(out, torch.bfloat16, "out")
    _req(residual_in, torch.bfloat16, "residual_in")
    rows, hidden = out.shape
    _native.check(_native.load().crag_enc_rmsnorm_partials(_p(partial_rows), int(splitk), int(m_pad), _p(residual_in), _p(weight),
                                                           _p(out), _p(residual_out), int(rows), int(hidden), float(eps),
                                                           _stream()), "crag_enc_rmsnorm_partials")
    return out
