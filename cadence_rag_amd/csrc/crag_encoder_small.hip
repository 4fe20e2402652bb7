// crag_encoder_small.hip — the Qwen3-Embedding decoder layer at the reference's own operating point: ONE short
// query per /retrieve request (/root/reference/app/retrieve.py:427 embeds one string), i.e. 16 or 32 token rows.
// C ABI: include/crag_encoder.h (crag_enc_small_gemm, crag_enc_small_attention; crag_enc_small_attention_seqs runs the
// attention kernel over a packed batch of short sequences, one workgroup per (q head, sequence)).
//
// At that size a layer is a WEIGHT STREAM: 202 MB of bf16 weights against 16 rows of activations, 25 us at
// 8 TB/s.  Round 3's first version ran a layer as eight launches (two RMSNorms, q/k-norm + RoPE + V transpose,
// the flash-attention kernel built for 256-token chunks, four weight-streaming GEMMs): 77 us of kernel time per
// layer, 2.8 ms per query.  This file is the layer as FIVE launches, each built for 16-32 rows:
//
//   small_gemm<PRO = 1>      resid' = resid + delta;  x = RMSNorm(resid') * w;  qkv = x W_qkv^T
//   small_attn               per-head q/k RMSNorm + RoPE + causal attention of <= 32 tokens, one workgroup per q head
//   small_gemm               delta = attn W_o^T
//   small_gemm<PRO = 1, EPI> resid'' = resid' + delta;  x = RMSNorm(resid'') * w;  act = silu(x W_g^T) * (x W_u^T)
//   small_gemm               delta = act W_d^T
//
// What changed against crag_enc_skinny_gemm (crag_encoder.hip), and why:
//   * the residual add and the RMSNorm run in the PROLOGUE of the projection that consumes them, on the B
//     fragments in registers, behind the first weight loads (every workgroup repeats them for the 16 rows: 160 KB
//     of L2 reads, while its first 80 KB of weights are on their way from HBM) -- two launches per layer less;
//   * the n-tile height R follows the chip: R rows of the weight per tile with R * tiles = N and tiles a multiple
//     of the 256 CUs wherever N allows -- R = 10 for N = 2560 (o, down: 256 tiles, one per CU; with 16-row tiles
//     160 workgroups streamed 304 KB each while 96 CUs idled: the whole forward 2.53 vs 2.57 ms with 16-row tiles
//     for down, 3.60 vs 3.68 ms at 32 rows), R = 12 for qkv (512 tiles), 16 for gate|up (1216);
//     an MFMA still multiplies 16 rows, the lanes of the missing rows re-read the tile's last row (same cache
//     line, no HBM bytes) and their results are dropped;
//   * one resident workgroup per CU walks its tiles (tile, tile + grid, ...) with the weight ring running ONE TILE
//     AHEAD across the tile boundary, the split-K reduction double-buffered in LDS: one barrier per tile, the
//     stream never drains inside a launch;
//   * attention for <= 32 tokens is a different problem from 256-token chunks: no K/V tiles to stream, the whole
//     head fits one wave's MFMAs (S^T = K Q^T: 4 MFMAs 16x16x32, P V: 8), and the time is the dependent chain
//     load -> norm -> RoPE -> LDS -> MFMA -> softmax -> MFMA -> store; q/k-norm + RoPE are fused in front (the
//     K head is normalised once per q head: 4x redundant, 16 x 128 elements).
//
// Arithmetic: the same roundings as the kernels these replace (bf16 residual add, fp32 statistics, bf16(x * rstd)
// then * w rounded to bf16; cos/sin cast to bf16; P rounded to bf16 after an fp32 softmax) -- tests compare the
// fused path with the unfused kernels and with transformers' Qwen3Model.

#include "crag_arch.h"
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "../../include/crag_encoder.h"

extern "C" void crag_set_error_(const char *msg);  // crag_api.hip

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint16_t u16;

int efail(const char *fmt, ...) {
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    crag_set_error_(buf);
    return -1;
}

int hip_ok(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s launch failed: %s", what, hipGetErrorString(e));
        crag_set_error_(buf);
        return -2;
    }
    return 0;
}

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }

struct SmallGemmParams {
    const u16 *x;       // [16 * MG, K] bf16 activations (PRO: the residual stream)
    const u16 *delta;   // PRO: [16 * MG, K], added to x in bf16
    const u16 *norm_w;  // PRO: [K]
    u16 *res_out;       // PRO: x + delta (written by workgroup 0), may be NULL
    const u16 *wsw;     // weights, [N / R][K / 32][4][R][8]
    u16 *out;           // [m_rows, ld_out]
    int m_rows, tiles, ld_out;
    float eps;
};

// MG token groups of 16, K = WAVES * KS * 32 split over the waves, D weight loads (D KiB at R = 16) in flight per
// wave, R weight rows per n-tile, EPI 1 = SwiGLU (R = 16: 8 gate rows then the 8 up rows of the same features),
// PRO 1 = residual add + RMSNorm in front, MULTI = a workgroup walks several tiles (needs KS % D == 0: the slot a
// k-step frees is the slot the same k-step of the next tile uses), XPASS = the B fragments are loaded in XPASS
// parts (K = 9728 at 32 rows: 304 registers otherwise).
template <int MG, int KS, int WAVES, int D, int R, int EPI, int PRO, int MULTI, int XPASS>
__global__ __launch_bounds__(WAVES * 64) void small_gemm_kernel(SmallGemmParams p) {
    static_assert(D <= KS && (!MULTI || KS % D == 0), "ring depth");
    static_assert(KS % XPASS == 0 && (XPASS == 1 || (!PRO && !MULTI)), "x passes");
    static_assert(EPI == 0 || R == 16, "the SwiGLU tile is 8 gate + 8 up rows");
    static_assert((R & 1) == 0 && R <= 16, "tile height");
    constexpr int KSTEPS = WAVES * KS, K = KSTEPS * 32, XS = KS / XPASS;
    __shared__ f32x4_t red[2][WAVES][MG][64];
    __shared__ float nrm[PRO ? WAVES : 1][MG][16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = lane >> 4, c = lane & 15;
    const int kw0 = w * KS;
    int tile = blockIdx.x;
    if (tile >= p.tiles) return;

    // B fragments: B[k = 8 g + j][col = c] = x[token 16 mg + c][32 (kw0 + s) + 8 g + j]
    bf16x8 xb[MG][XS];
    bf16x8 nw[PRO ? KS : 1];
    auto load_x = [&](int pass) {
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            const u16 *xr = p.x + (size_t)(16 * mg + c) * K + (size_t)(kw0 + pass * XS) * 32 + 8 * g;
#pragma unroll
            for (int s = 0; s < XS; ++s) xb[mg][s] = *reinterpret_cast<const bf16x8 *>(xr + 32 * s);
        }
    };
    // weights: a (tile, k-step) block is 4 k-groups x R rows x 8 elements; lanes of the rows a short tile lacks
    // re-read its last row.  The activation loads go first (L2; with the prologue the norm is computed while the
    // weights are on their way from HBM).  [Measured: the first D weight loads in front of the activation loads,
    // 7.5 -> 8.4 us for the o projection, no change for down.]
    const int crow = c < R ? c : R - 1;
    const u16 *wlane = p.wsw + (size_t)kw0 * (32 * R) + (size_t)(g * R + crow) * 8;
    auto wptr = [&](int t, int s) -> const bf16x8 * {
        return reinterpret_cast<const bf16x8 *>(wlane + ((size_t)t * KSTEPS + s) * (32 * R));
    };
    load_x(0);
    bf16x8 dl[PRO ? MG : 1][PRO ? KS : 1];
    if constexpr (PRO) {
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            const u16 *dr = p.delta + (size_t)(16 * mg + c) * K + (size_t)kw0 * 32 + 8 * g;
#pragma unroll
            for (int s = 0; s < KS; ++s) dl[mg][s] = *reinterpret_cast<const bf16x8 *>(dr + 32 * s);
        }
        if constexpr (MG == 1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) nw[s] = *reinterpret_cast<const bf16x8 *>(p.norm_w + (size_t)(kw0 + s) * 32 + 8 * g);
        }
    }
    bf16x8 wr[D];
    if constexpr (!(PRO && MG > 1)) {
#pragma unroll
        for (int i = 0; i < D; ++i) wr[i] = __builtin_nontemporal_load(wptr(tile, i));
    }

    if constexpr (PRO) {
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            const int token = 16 * mg + c;
            float ss = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 a = xb[mg][s];
                const bf16x8 d = dl[mg][s];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const u16 sv = f2bf(bf2f((u16)a[e]) + bf2f((u16)d[e]));  // bf16 add, as the model does
                    const float v = bf2f(sv);
                    a[e] = (short)sv;
                    ss += v * v;
                }
                xb[mg][s] = a;
                if (p.res_out && blockIdx.x == 0 && token < p.m_rows)
                    *reinterpret_cast<bf16x8 *>(p.res_out + (size_t)token * K + (size_t)(kw0 + s) * 32 + 8 * g) = a;
            }
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (lane < 16) nrm[w][mg][c] = ss;
        }
        if constexpr (MG > 1) {  // 32 rows: x and delta fill the register file; the ring starts when delta is dead
#pragma unroll
            for (int i = 0; i < D; ++i) wr[i] = __builtin_nontemporal_load(wptr(tile, i));
        }
        __syncthreads();
        if constexpr (MG > 1) {  // ... and the norm weights come last
#pragma unroll
            for (int s = 0; s < KS; ++s) nw[s] = *reinterpret_cast<const bf16x8 *>(p.norm_w + (size_t)(kw0 + s) * 32 + 8 * g);
        }
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            float tot = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) tot += nrm[ww][mg][c];
            const float rstd = rsqrtf(tot / (float)K + p.eps);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 a = xb[mg][s];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    a[e] = (short)f2bf(bf2f((u16)nw[s][e]) * bf2f(f2bf(bf2f((u16)a[e]) * rstd)));
                xb[mg][s] = a;
            }
        }
    }

    int buf = 0;
    for (; tile < p.tiles; tile += gridDim.x) {
        const int next = tile + (int)gridDim.x;
        f32x4_t acc[MG];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) acc[mg] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto kloop = [&](auto has_next_c) {
            constexpr bool HN = decltype(has_next_c)::value;
#pragma unroll
            for (int pass = 0; pass < XPASS; ++pass) {
                if (pass > 0) load_x(pass);
#pragma unroll
                for (int sx = 0; sx < XS; ++sx) {
                    const int s = pass * XS + sx;
                    const bf16x8 wv = wr[s % D];
                    if (s + D < KS) wr[s % D] = __builtin_nontemporal_load(wptr(tile, s + D));
                    else if (HN) wr[s % D] = __builtin_nontemporal_load(wptr(next, s + D - KS));
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg)
                        acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, xb[mg][sx], acc[mg], 0, 0, 0);
                }
            }
        };
        if (MULTI && next < p.tiles) kloop(std::true_type{});
        else kloop(std::false_type{});
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) red[buf][w][mg][lane] = acc[mg];
        __syncthreads();
        // D[row = 4 g + r][col = c]: lane (g, c) of wave mg owns 4 consecutive weight rows of token 16 mg + c
        if (w < MG) {
            const int mg = w;
            const int token = 16 * mg + c;
            f32x4_t sum = red[buf][0][mg][lane];
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) sum += red[buf][ww][mg][lane];
            if (EPI == 0) {
                if (4 * g < R && token < p.m_rows) {
                    uint32_t *o = reinterpret_cast<uint32_t *>(p.out + (size_t)token * p.ld_out + (size_t)R * tile + 4 * g);
                    o[0] = (uint32_t)f2bf(sum[0]) | ((uint32_t)f2bf(sum[1]) << 16);
                    if (4 * g + 2 < R) o[1] = (uint32_t)f2bf(sum[2]) | ((uint32_t)f2bf(sum[3]) << 16);
                }
            } else if (g < 2) {
                f32x4_t up = red[buf][0][mg][lane + 32];
#pragma unroll
                for (int ww = 1; ww < WAVES; ++ww) up += red[buf][ww][mg][lane + 32];
                if (token < p.m_rows) {
                    u16 o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gt = bf2f(f2bf(sum[r]));
                        const float act = bf2f(f2bf(gt / (1.f + __expf(-gt))));
                        o[r] = f2bf(act * bf2f(f2bf(up[r])));
                    }
                    *reinterpret_cast<uint2 *>(p.out + (size_t)token * p.ld_out + (size_t)8 * tile + 4 * g) =
                        make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
                }
            }
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------
// attention of <= 16 QB tokens: one workgroup per q head
// ---------------------------------------------------------------------------------------------
struct SmallAttnParams {
    const u16 *qkv;  // [T, (hq + 2 hkv) * 128]: projections BEFORE q/k-norm and RoPE
    const u16 *qw, *kw;
    const float *cos_sin;  // [max_pos, 64, 2], or (cs_by_token) [T, 64, 2]: the rows of the tokens' positions
    const int32_t *positions;
    u16 *out;  // [T, hq * 128]
    const int32_t *cu;  // NULL: ONE block of n_tokens rows; else [gridDim.y + 1]: block y = rows cu[y] .. cu[y + 1] - 1 (one sequence)
    const float *parts; // nullable: the qkv projection as split-K partial tiles [splitk][m_pad][width] fp32 (token-major) instead
    int splitk, m_pad;  //           of qkv: a head vector = bf16(sum over the splits, in split order)
    int n_tokens, hq, hkv, cs_by_token;
    float eps, scale_log2;
};

struct alignas(16) Pack8s {
    u16 v[8];
};

template <int QB>
__global__ __launch_bounds__(256 * QB) void small_attn_kernel(SmallAttnParams p) {
    constexpr int T = 16 * QB;
    constexpr int THREADS = 256 * QB, GROUPS = THREADS / 16;
    constexpr int ITEMS = 3 * T / GROUPS;  // (token, q | k | v) head vectors per 16-lane group: three, for 16 and 32 rows
    __shared__ alignas(16) u16 Qs[T][136];
    __shared__ alignas(16) u16 Ks[T][136];
    __shared__ alignas(16) u16 Vt[CRAG_HEAD_DIM][40];  // [d][key slot in PV-fragment order], 32 slots + pad
    __shared__ int start[T];
    const int h = blockIdx.x;
    const int kvh = h / (p.hq / p.hkv);
    const int64_t row_stride = (int64_t)(p.hq + 2 * p.hkv) * CRAG_HEAD_DIM;
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int tok0 = p.cu ? p.cu[blockIdx.y] : 0;
    const int ntok = p.cu ? min(p.cu[blockIdx.y + 1] - tok0, T) : p.n_tokens;

    Pack8s raw[ITEMS];
    float cs[ITEMS][16];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int it = grp + GROUPS * i, t = it / 3, which = it % 3;
        const int head = which == 0 ? h : (which == 1 ? p.hq + kvh : p.hq + p.hkv + kvh);
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[i].v[e] = 0;
        if (t < ntok) {
            if (p.parts) {   // the projection's split-K partial tiles: summed and rounded here (no reduce launch)
                f32x4_t lo = f32x4_t{0.f, 0.f, 0.f, 0.f}, hi = lo;
                for (int s = 0; s < p.splitk; ++s) {
                    const float *src = p.parts + ((int64_t)s * p.m_pad + tok0 + t) * row_stride + (int64_t)head * CRAG_HEAD_DIM + sub * 8;
                    lo += *reinterpret_cast<const f32x4_t *>(src);
                    hi += *reinterpret_cast<const f32x4_t *>(src + 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    raw[i].v[e] = f2bf(lo[e]);
                    raw[i].v[4 + e] = f2bf(hi[e]);
                }
            } else
            raw[i] = *reinterpret_cast<const Pack8s *>(p.qkv + (tok0 + t) * row_stride + (int64_t)head * CRAG_HEAD_DIM + sub * 8);
            if (which != 2) {
                // a table gathered per token (once per forward) takes a dependent load out of every layer's chain
                const float *src = p.cos_sin + ((int64_t)(p.cs_by_token ? tok0 + t : p.positions[tok0 + t]) * 64 + (sub & 7) * 8) * 2;
#pragma unroll
                for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4 *>(&cs[i][e]) = *reinterpret_cast<const float4 *>(src + e);
            }
        }
    }
    const Pack8s wq8 = *reinterpret_cast<const Pack8s *>(p.qw + sub * 8);
    const Pack8s wk8 = *reinterpret_cast<const Pack8s *>(p.kw + sub * 8);
    for (int i = threadIdx.x; i < CRAG_HEAD_DIM * 40 / 2; i += THREADS) reinterpret_cast<uint32_t *>(&Vt[0][0])[i] = 0;
    if (threadIdx.x < T)
        start[threadIdx.x] = (int)threadIdx.x < ntok ? (int)threadIdx.x - p.positions[tok0 + threadIdx.x] : -1 - (int)threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int it = grp + GROUPS * i, t = it / 3, which = it % 3;
        if (which == 2) {
            const int slot = 8 * ((t & 15) >> 2) + 4 * (t >> 4) + (t & 3);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[sub * 8 + e][slot] = raw[i].v[e];
            continue;
        }
        // per-head RMSNorm + rotate-half RoPE: the arithmetic of qk_norm_rope_body (crag_encoder.hip)
        const Pack8s &w8 = which == 0 ? wq8 : wk8;
        float v[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = bf2f(raw[i].v[e]);
            ss += v[e] * v[e];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float rstd = rsqrtf(ss / (float)CRAG_HEAD_DIM + p.eps);
        Pack8s o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float n = bf2f(f2bf(bf2f(w8.v[e]) * bf2f(f2bf(v[e] * rstd))));
            const float partner = __shfl_xor(n, 8);
            const float rot = sub < 8 ? -partner : partner;
            const float cv = bf2f(f2bf(cs[i][2 * e])), sn = bf2f(f2bf(cs[i][2 * e + 1]));  // the model casts cos/sin to bf16
            o8.v[e] = t < ntok ? f2bf(n * cv + rot * sn) : (u16)0;
        }
        *reinterpret_cast<Pack8s *>(which == 0 ? &Qs[t][sub * 8] : &Ks[t][sub * 8]) = o8;
    }
    __syncthreads();

    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= QB) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int qtok = 16 * w + c;
    // S^T block kb: D[row = key 16 kb + 4 g + r][col = query 16 w + c]
    f32x4_t st[QB];
#pragma unroll
    for (int kb = 0; kb < QB; ++kb) {
        st[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (kb <= w) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&Ks[16 * kb + c][32 * s + 8 * g]);
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Qs[qtok][32 * s + 8 * g]);
                st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, st[kb], 0, 0, 0);
            }
        }
    }
    const int qs = start[qtok];
    bool ok[QB][4];
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ktok = 16 * kb + 4 * g + r;
            ok[kb][r] = kb <= w && ktok <= qtok && start[ktok] == qs;
            if (ok[kb][r]) m = fmaxf(m, st[kb][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float pr[QB][4], l = 0.f;
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pr[kb][r] = ok[kb][r] ? exp2f((st[kb][r] - m) * p.scale_log2) : 0.f;
            l += pr[kb][r];
        }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    // P^T as the B operand: MFMA k index 8 g + j <-> key 4 g + j (block 0, j < 4), 16 + 4 g + j - 4 (block 1):
    // exactly this lane's S^T registers; Vt stores the keys in that order
    bf16x8 pv;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = (j >> 2) < QB ? (short)f2bf(pr[(j >> 2) < QB ? (j >> 2) : 0][j & 3] * inv) : (short)0;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&Vt[16 * nt + c][8 * g]);
        const f32x4_t o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pv, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // O^T[d = 16 nt + 4 g + r][query c]
        if (qtok < ntok)
            *reinterpret_cast<uint2 *>(p.out + (size_t)(tok0 + qtok) * p.hq * CRAG_HEAD_DIM + (size_t)h * CRAG_HEAD_DIM + 16 * nt + 4 * g) =
                make_uint2((uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16), (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16));
    }
}

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

}  // namespace

extern "C" {

int crag_enc_small_gemm(const uint16_t *x, const uint16_t *delta, const uint16_t *norm_w, uint16_t *res_out,
                        const uint16_t *wsw, uint16_t *out, int m_rows, int m_pad, int n, int k, int rows_per_tile,
                        int epilogue, float eps, void *stream) {
    if (!x || !wsw || !out) return efail("small_gemm: NULL pointer");
    if (m_rows <= 0 || m_rows > m_pad || (m_pad != 16 && m_pad != 32))
        return efail("small_gemm: m_pad must be 16 or 32 and 0 < m_rows <= m_pad");
    if (epilogue != 0 && epilogue != 1) return efail("small_gemm: epilogue must be 0 or 1");
    const bool pro = norm_w != nullptr;
    if (pro && !delta) return efail("small_gemm: the RMSNorm prologue needs a delta (pass zeros for none)");
    if (!pro && (delta || res_out)) return efail("small_gemm: delta / res_out only with the RMSNorm prologue");
    if (rows_per_tile <= 0 || n <= 0 || n % rows_per_tile) return efail("small_gemm: n must be a multiple of rows_per_tile");
    SmallGemmParams p;
    p.x = x;
    p.delta = delta;
    p.norm_w = norm_w;
    p.res_out = res_out;
    p.wsw = wsw;
    p.out = out;
    p.m_rows = m_rows;
    p.tiles = n / rows_per_tile;
    p.ld_out = epilogue ? n / 2 : n;
    p.eps = eps;
    const int mg = m_pad / 16;
    const int cus = cu_count();
    hipStream_t st = (hipStream_t)stream;
#define CRAG_SMALL(MG_, KS_, WAVES_, D_, R_, EPI_, PRO_, MULTI_, XP_)                                              \
    hipLaunchKernelGGL((small_gemm_kernel<MG_, KS_, WAVES_, D_, R_, EPI_, PRO_, MULTI_, XP_>),                      \
                       dim3((unsigned)((MULTI_) && p.tiles > cus ? cus : p.tiles)), dim3(WAVES_ * 64), 0, st, p)
    if (k == 2560 && pro && epilogue == 0 && rows_per_tile == 12) {
        if (mg == 1) CRAG_SMALL(1, 10, 8, 10, 12, 0, 1, 1, 1); else CRAG_SMALL(2, 10, 8, 10, 12, 0, 1, 1, 1);
    } else if (k == 2560 && pro && epilogue == 1 && rows_per_tile == 16) {
        if (mg == 1) CRAG_SMALL(1, 10, 8, 10, 16, 1, 1, 1, 1); else CRAG_SMALL(2, 10, 8, 10, 16, 1, 1, 1, 1);
    } else if (k == 4096 && !pro && epilogue == 0 && rows_per_tile == 10) {
        if (mg == 1) CRAG_SMALL(1, 16, 8, 16, 10, 0, 0, 0, 1); else CRAG_SMALL(2, 16, 8, 16, 10, 0, 0, 0, 1);
    } else if (k == 9728 && !pro && epilogue == 0 && rows_per_tile == 10) {
        // 16 rows: ring depth 19 = half of a wave's 38 k-steps (2.528 -> 2.477 ms per one-query forward against depth
        // 12; depth 16: 2.523; the whole tile up front with the activations in two passes: 2.510 --
        // profiles/r04_nq1_variants.txt; there too: both qkv tiles of a workgroup up front, no gain)
        if (mg == 1) CRAG_SMALL(1, 38, 8, 19, 10, 0, 0, 0, 1); else CRAG_SMALL(2, 38, 8, 12, 10, 0, 0, 0, 2);
    } else {
        return efail("small_gemm: unsupported form k=%d rows_per_tile=%d epilogue=%d prologue=%d (built for the "
                     "Qwen3-Embedding-4B widths: k 2560 with the RMSNorm prologue and 12 / 16-row tiles, k 4096 and "
                     "9728 with 10-row tiles)", k, rows_per_tile, epilogue, (int)pro);
    }
#undef CRAG_SMALL
    return hip_ok("small_gemm");
}

int crag_enc_small_attention(const uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w,
                             const float *cos_sin, int cos_sin_by_token, const int32_t *positions, uint16_t *out,
                             int n_tokens, int hq, int hkv, float eps, float scale, void *stream) {
    if (!qkv || !q_norm_w || !k_norm_w || !cos_sin || !positions || !out) return efail("small_attention: NULL pointer");
    if (n_tokens <= 0 || n_tokens > 32) return efail("small_attention: 1..32 tokens (got %d)", n_tokens);
    if (hq <= 0 || hkv <= 0 || hq % hkv) return efail("small_attention: hq must be a multiple of hkv");
    SmallAttnParams p;
    p.qkv = qkv;
    p.qw = q_norm_w;
    p.kw = k_norm_w;
    p.cos_sin = cos_sin;
    p.positions = positions;
    p.out = out;
    p.n_tokens = n_tokens;
    p.hq = hq;
    p.hkv = hkv;
    p.cu = nullptr;
    p.parts = nullptr;
    p.splitk = p.m_pad = 0;
    p.cs_by_token = cos_sin_by_token != 0;
    p.eps = eps;
    p.scale_log2 = scale * 1.4426950408889634f;
    if (n_tokens <= 16) hipLaunchKernelGGL(small_attn_kernel<1>, dim3((unsigned)hq), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(small_attn_kernel<2>, dim3((unsigned)hq), dim3(512), 0, (hipStream_t)stream, p);
    return hip_ok("small_attention");
}

static int small_attention_seqs_impl(const uint16_t *qkv, const float *parts, int splitk, int m_pad, const uint16_t *q_norm_w,
                                     const uint16_t *k_norm_w, const float *cos_sin, int cos_sin_by_token,
                                     const int32_t *positions, const int32_t *cu_seqlens, int n_seqs, int max_len,
                                     uint16_t *out, int hq, int hkv, float eps, float scale, void *stream);

int crag_enc_small_attention_seqs(const uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w,
                                  const float *cos_sin, int cos_sin_by_token, const int32_t *positions,
                                  const int32_t *cu_seqlens, int n_seqs, int max_len, uint16_t *out, int hq, int hkv,
                                  float eps, float scale, void *stream) {
    if (!qkv) return efail("small_attention_seqs: NULL pointer");
    return small_attention_seqs_impl(qkv, nullptr, 0, 0, q_norm_w, k_norm_w, cos_sin, cos_sin_by_token, positions, cu_seqlens,
                                     n_seqs, max_len, out, hq, hkv, eps, scale, stream);
}

int crag_enc_small_attention_seqs_parts(const float *qkv_partial_rows, int splitk, int m_pad, const uint16_t *q_norm_w,
                                        const uint16_t *k_norm_w, const float *cos_sin, int cos_sin_by_token,
                                        const int32_t *positions, const int32_t *cu_seqlens, int n_seqs, int max_len,
                                        uint16_t *out, int hq, int hkv, float eps, float scale, void *stream) {
    if (!qkv_partial_rows || splitk <= 0 || m_pad <= 0) return efail("small_attention_seqs_parts: NULL pointer / bad split");
    return small_attention_seqs_impl(nullptr, qkv_partial_rows, splitk, m_pad, q_norm_w, k_norm_w, cos_sin, cos_sin_by_token,
                                     positions, cu_seqlens, n_seqs, max_len, out, hq, hkv, eps, scale, stream);
}

static int small_attention_seqs_impl(const uint16_t *qkv, const float *parts, int splitk, int m_pad, const uint16_t *q_norm_w,
                                     const uint16_t *k_norm_w, const float *cos_sin, int cos_sin_by_token,
                                     const int32_t *positions, const int32_t *cu_seqlens, int n_seqs, int max_len,
                                     uint16_t *out, int hq, int hkv, float eps, float scale, void *stream) {
    if ((!qkv && !parts) || !q_norm_w || !k_norm_w || !cos_sin || !positions || !cu_seqlens || !out)
        return efail("small_attention_seqs: NULL pointer");
    if (n_seqs <= 0 || n_seqs > 65535) return efail("small_attention_seqs: 1..65535 sequences (got %d)", n_seqs);
    if (max_len <= 0 || max_len > 32) return efail("small_attention_seqs: sequences of 1..32 tokens (max_len %d)", max_len);
    if (hq <= 0 || hkv <= 0 || hq % hkv) return efail("small_attention_seqs: hq must be a multiple of hkv");
    SmallAttnParams p;
    p.qkv = qkv;
    p.qw = q_norm_w;
    p.kw = k_norm_w;
    p.cos_sin = cos_sin;
    p.positions = positions;
    p.out = out;
    p.cu = cu_seqlens;
    p.parts = parts;
    p.splitk = splitk;
    p.m_pad = m_pad;
    p.n_tokens = 0;
    p.hq = hq;
    p.hkv = hkv;
    p.cs_by_token = cos_sin_by_token != 0;
    p.eps = eps;
    p.scale_log2 = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)hq, (unsigned)n_seqs);
    if (max_len <= 16) hipLaunchKernelGGL(small_attn_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(small_attn_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, p);
    return hip_ok("small_attention_seqs");
}

}  // extern "C"
