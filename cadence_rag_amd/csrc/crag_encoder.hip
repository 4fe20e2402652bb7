// crag_encoder.hip — hand-written HIP operators of the Qwen3-Embedding encoder lane for gfx950.
// C ABI: include/crag_encoder.h.  Everything of the decoder forward except the plain linear layers
// (library GEMMs issued by the Python host) lives here: embedding gather, RMSNorm (+ residual),
// per-head q/k RMSNorm + RoPE, V transpose, causal GQA flash attention (bf16 MFMA), SwiGLU, and the
// gateway's pooling / slice / L2 normalisation.
//
// Reference math (not code): P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:683-716 for the
// post-processing; the model family's public architecture for the layer (checked against
// transformers' Qwen3Model in tests/test_encoder_gpu.py).

#include "crag_arch.h"
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/crag_encoder.h"

extern "C" const char *crag_last_error(void);
extern "C" void crag_set_error_(const char *msg);  // defined in crag_api.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ u16 f2bf(float f) {  // round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(u16, (__bf16)f);
}

struct alignas(16) Pack8 {
    u16 v[8];
};

// ---------------------------------------------------------------------------------------------
// embedding gather
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_gather_kernel(const int32_t *ids, const u16 *table, u16 *out,
                                                           int64_t n_tokens, int hidden, int64_t vocab) {
    const int chunks = hidden >> 3;
    for (int64_t t = blockIdx.x; t < n_tokens; t += gridDim.x) {
        int64_t id = ids[t];
        if (id < 0) id = 0;
        if (id >= vocab) id = vocab - 1;
        const Pack8 *src = reinterpret_cast<const Pack8 *>(table + id * hidden);
        Pack8 *dst = reinterpret_cast<Pack8 *>(out + t * hidden);
        for (int c = threadIdx.x; c < chunks; c += blockDim.x) dst[c] = src[c];
    }
}

// ---------------------------------------------------------------------------------------------
// RMSNorm (+ residual add)
// ---------------------------------------------------------------------------------------------
constexpr int NORM_THREADS = 256;
constexpr int NORM_MAX_CHUNKS = 4;  // hidden <= 8 * 256 * 4 = 8192

__device__ __forceinline__ float block_sum(float v, float *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wv] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_kernel(const u16 *x, const u16 *res_in, const u16 *w,
                                                               u16 *out, u16 *res_out, int64_t rows, int hidden,
                                                               float eps) {
    __shared__ float sh[NORM_THREADS / 64];
    const int chunks = hidden >> 3;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const Pack8 *xr = reinterpret_cast<const Pack8 *>(x + r * hidden);
        const Pack8 *rr = res_in ? reinterpret_cast<const Pack8 *>(res_in + r * hidden) : nullptr;
        float v[NORM_MAX_CHUNKS][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
            const int c = threadIdx.x + i * NORM_THREADS;
            if (c < chunks) {
                const Pack8 a = xr[c];
                Pack8 s8;
                if (rr) {
                    const Pack8 b = rr[c];
#pragma unroll
                    for (int e = 0; e < 8; ++e) s8.v[e] = f2bf(bf2f(a.v[e]) + bf2f(b.v[e]));  // bf16 add, as the model does
                } else {
                    s8 = a;
                }
                if (res_out) reinterpret_cast<Pack8 *>(res_out + r * hidden)[c] = s8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[i][e] = bf2f(s8.v[e]);
                    ss += v[i][e] * v[i][e];
                }
            }
        }
        ss = block_sum(ss, sh);
        const float rstd = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
        for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
            const int c = threadIdx.x + i * NORM_THREADS;
            if (c < chunks) {
                const Pack8 w8 = reinterpret_cast<const Pack8 *>(w)[c];
                Pack8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[e] = f2bf(bf2f(w8.v[e]) * bf2f(f2bf(v[i][e] * rstd)));
                reinterpret_cast<Pack8 *>(out + r * hidden)[c] = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-head q/k RMSNorm + RoPE, in place on the fused qkv rows
// 16 lanes per head vector (8 elements each); lane j of the group pairs with lane j^8 for
// rotate_half (element i <-> i + 64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void qk_norm_rope_body(u16 *qkv, const u16 *qw, const u16 *kw, const float *cos_sin,
                                                  const int32_t *positions, int64_t n_tokens, int hq, int hkv,
                                                  float eps, int64_t first_token, int64_t token_stride) {
    // One block trip = one token: group g (16 lanes) walks the head vectors g, g + 16, g + 32, ... of the token with
    // the token's cos/sin chunk and the norm weights in registers (loaded per head vector they were 64 B of table
    // reads per 16 B of q/k: twice the data traffic, from the caches).
    const int heads = hq + hkv;  // q heads then k heads are contiguous in the row
    const int64_t row_stride = (int64_t)(hq + 2 * hkv) * CRAG_HEAD_DIM;
    const int g = threadIdx.x >> 4;
    const int sub = threadIdx.x & 15;  // 8-element chunk of the head
    const bool first_half = sub < 8;   // elements [0, 64): out = x*cos - x[i+64]*sin ; else x*cos + x[i-64]*sin
    const Pack8 wq8 = *reinterpret_cast<const Pack8 *>(qw + sub * 8);
    const Pack8 wk8 = *reinterpret_cast<const Pack8 *>(kw + sub * 8);
    for (int64_t t = first_token; t < n_tokens; t += token_stride) {
        const int pos = positions[t];
        const float *cs = cos_sin + ((int64_t)pos * 64 + (sub & 7) * 8) * 2;
        float c[8], sn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {  // the model casts cos/sin to bf16
            c[e] = bf2f(f2bf(cs[2 * e]));
            sn[e] = bf2f(f2bf(cs[2 * e + 1]));
        }
        for (int hd = g; hd < heads; hd += 16) {
            u16 *p = qkv + t * row_stride + (int64_t)hd * CRAG_HEAD_DIM + sub * 8;
            const Pack8 a = *reinterpret_cast<Pack8 *>(p);
            const Pack8 &w8 = hd < hq ? wq8 : wk8;
            float v[8];
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = bf2f(a.v[e]);
                ss += v[e] * v[e];
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);  // 16-lane group
            const float rstd = rsqrtf(ss / (float)CRAG_HEAD_DIM + eps);
            float n[8], partner[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) n[e] = bf2f(f2bf(bf2f(w8.v[e]) * bf2f(f2bf(v[e] * rstd))));
#pragma unroll
            for (int e = 0; e < 8; ++e) partner[e] = __shfl_xor(n[e], 8);  // rotate_half: element i <-> i + 64
            Pack8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float rot = first_half ? -partner[e] : partner[e];
                o.v[e] = f2bf(n[e] * c[e] + rot * sn[e]);
            }
            *reinterpret_cast<Pack8 *>(p) = o;
        }
    }
}

__global__ __launch_bounds__(256) void qk_norm_rope_kernel(u16 *qkv, const u16 *qw, const u16 *kw,
                                                           const float *cos_sin, const int32_t *positions,
                                                           int64_t n_tokens, int hq, int hkv, float eps) {
    qk_norm_rope_body(qkv, qw, kw, cos_sin, positions, n_tokens, hq, hkv, eps, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// V transpose: qkv[T, ..] V part -> Vt[hkv][128][t_pad]; one block per (32 padded slots, kv head)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void v_transpose_body(const u16 *qkv, u16 *vt, const int32_t *tok_of_pad, int64_t t_pad,
                                                 int hq, int hkv, int slot_block, int kvh_,
                                                 u16 (*tile)[CRAG_HEAD_DIM + 8]) {
    const int64_t p0 = (int64_t)slot_block * 32;
    const int kvh = kvh_;
    const int64_t row_stride = (int64_t)(hq + 2 * hkv) * CRAG_HEAD_DIM;
    {
        const int slot = threadIdx.x >> 3, c8 = threadIdx.x & 7;  // 32 slots x 8 chunks of 16 elements
        const int32_t tok = (p0 + slot < t_pad) ? tok_of_pad[p0 + slot] : -1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int d0 = c8 * 16 + half * 8;
            Pack8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a.v[e] = 0;
            if (tok >= 0)
                a = *reinterpret_cast<const Pack8 *>(qkv + (int64_t)tok * row_stride +
                                                     (int64_t)(hq + hkv + kvh) * CRAG_HEAD_DIM + d0);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[slot][d0 + e] = a.v[e];
        }
    }
    __syncthreads();
    {
        // each thread writes 16 slots (32 B) of one d row: 128 d rows x 2 halves = 256 threads.
        // Inside a 32-slot block the slots are stored in PV-fragment order (see attention_kernel): stored
        // index 16 s2 + 8 h + 4 g + r holds slot 16 s2 + 8 g + 4 h + r, so that the 8 keys one lane feeds
        // to one PV MFMA are 16 contiguous bytes.
        const int d = threadIdx.x >> 1, half = threadIdx.x & 1;
        u16 *dst = vt + ((int64_t)kvh * CRAG_HEAD_DIM + d) * t_pad + p0 + half * 16;
        Pack8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o0.v[e] = tile[half * 16 + 8 * (e >> 2) + (e & 3)][d];      // h = 0
            o1.v[e] = tile[half * 16 + 8 * (e >> 2) + 4 + (e & 3)][d];  // h = 1
        }
        reinterpret_cast<Pack8 *>(dst)[0] = o0;
        reinterpret_cast<Pack8 *>(dst)[1] = o1;
    }
}

__global__ __launch_bounds__(256) void v_transpose_kernel(const u16 *qkv, u16 *vt, const int32_t *tok_of_pad,
                                                          int64_t t_pad, int hq, int hkv) {
    __shared__ u16 tile[32][CRAG_HEAD_DIM + 8];
    v_transpose_body(qkv, vt, tok_of_pad, t_pad, hq, hkv, blockIdx.x, blockIdx.y, tile);
}

// Both in ONE launch (they touch disjoint columns of the fused qkv rows: q|k in place, v -> V^T): blocks
// [0, rope_blocks) walk the tokens, the rest are the (slot block, kv head) tiles of the transpose.  At 16 tokens a
// decoder layer is eight launches of a few microseconds each; every boundary removed is ~4 us of 80.
__global__ __launch_bounds__(256) void qk_rope_vt_kernel(u16 *qkv, const u16 *qw, const u16 *kw, const float *cos_sin,
                                                         const int32_t *positions, int64_t n_tokens, int hq, int hkv,
                                                         float eps, u16 *vt, const int32_t *tok_of_pad, int64_t t_pad,
                                                         int rope_blocks) {
    __shared__ u16 tile[32][CRAG_HEAD_DIM + 8];
    if ((int)blockIdx.x < rope_blocks) {
        qk_norm_rope_body(qkv, qw, kw, cos_sin, positions, n_tokens, hq, hkv, eps, blockIdx.x, rope_blocks);
    } else {
        const int b = (int)blockIdx.x - rope_blocks;
        v_transpose_body(qkv, vt, tok_of_pad, t_pad, hq, hkv, b / hkv, b % hkv, tile);
    }
}

// ---------------------------------------------------------------------------------------------
// causal GQA flash attention, head_dim 128, one wave = 32 query rows of one query head.
//   S^T = K . Q^T   (A = K tile rows, B = Q^T)   -> lane holds 16 of the 32 keys of ONE query row
//   O^T += V^T . P^T (A = V^T tile from the transposed copy, B = P^T taken straight from the S^T
//                     accumulator registers, bf16-packed; k order as the 32x32 C/D map gives it)
// ---------------------------------------------------------------------------------------------
struct AttnParams {
    const u16 *qkv;
    const u16 *vt;
    u16 *out;
    const int32_t *cu, *cu_pad, *blk_seq, *blk_q0;
    int64_t t_pad;
    int hq, hkv;
    float scale_log2;
};

__device__ __forceinline__ bf16x8 ld_frag(const u16 *p) { return *reinterpret_cast<const bf16x8 *>(p); }

// The 4 (hq/hkv) waves of a workgroup are the query heads of one GQA group: they need the SAME K and V
// tiles.  Loading fragments straight from global memory uses 32 B of every 128-B line per instruction
// and repeats the traffic per wave, which makes the kernel L1-bound; instead the workgroup stages each
// 32-key tile once, fully coalesced, in LDS (double-buffered, one barrier per tile) and the waves read
// their MFMA fragments from there (rows padded to 272 / 80 bytes: conflict-free ds_read_b128).
constexpr int ATT_KROW = 136;   // u16 per staged K row (128 + 8 pad)
constexpr int ATT_VROW = 40;    // u16 per staged V^T row (32 + 8 pad)

template <int GROUP>
__global__ __launch_bounds__(64 * GROUP) __attribute__((amdgpu_waves_per_eu(GROUP >= 2 ? 2 : 1, 8)))
void attention_kernel(AttnParams p) {
    // one pool: K buffers, then V^T buffers; the epilogue reuses its start for the per-wave output tiles
    constexpr int K_BUF = 32 * ATT_KROW, V_BUF = CRAG_HEAD_DIM * ATT_VROW;
    static_assert(GROUP * 32 * ATT_KROW <= 2 * (K_BUF + V_BUF) || GROUP > 4, "output tiles must fit the staging pool");
    __shared__ __attribute__((aligned(16))) u16 s_pool[2 * (K_BUF + V_BUF)];
    u16(*s_k)[K_BUF] = reinterpret_cast<u16(*)[K_BUF]>(s_pool);
    u16(*s_v)[V_BUF] = reinterpret_cast<u16(*)[V_BUF]>(s_pool + 2 * K_BUF);
    constexpr int nthr = 64 * GROUP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // kv head on the fast grid axis: workgroups go to the 8 XCDs round-robin by linear id, so with 8 kv heads
    // every XCD serves ONE kv head and the q blocks that re-read a sequence's K/V tiles share that XCD's L2
    const int kvh = blockIdx.x;
    const int head = kvh * GROUP + wave;  // GROUP = query heads per kv head = waves per workgroup
    const int seq = p.blk_seq[blockIdx.y];
    const int q0 = p.blk_q0[blockIdx.y];
    const int s_begin = p.cu[seq];
    const int len = p.cu[seq + 1] - s_begin;
    const int64_t pad_base = p.cu_pad[seq];
    const int c = lane & 31, h = lane >> 5;
    const int64_t row_stride = (int64_t)(p.hq + 2 * p.hkv) * CRAG_HEAD_DIM;

    // Q^T fragments (B operand): B[k = 8h + j][col c] = Q[q0 + c][16 s + 8h + j]
    bf16x8 qf[8];
    {
        const u16 *qp = p.qkv + (int64_t)(s_begin + q0 + c) * row_stride + (int64_t)head * CRAG_HEAD_DIM + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = ld_frag(qp + 16 * s);
    }
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 oacc[4] = {zero, zero, zero, zero};
    float m = -INFINITY, l = 0.f;
    const int n_kt = q0 / 32 + 1;
    const u16 *kglob = p.qkv + (int64_t)s_begin * row_stride + (int64_t)(p.hq + kvh) * CRAG_HEAD_DIM;
    const u16 *vglob = p.vt + (int64_t)kvh * CRAG_HEAD_DIM * p.t_pad + pad_base;

    // cooperative staging: the K tile is 32 rows x 16 chunks of 16 B, the V^T tile 128 rows x 4 chunks; with
    // `nthr` threads every thread moves 512 / nthr chunks of each (nthr = 64 * group, group in {1, 2, 4, 8})
    constexpr int per = 512 / nthr;
    struct Stage {
        bf16x8 k[per], v[per];
    };
    auto fetch = [&](int kt, Stage &st) {
        const int k0 = kt * 32;
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int ch = tid + i * nthr;
            st.k[i] = ld_frag(kglob + (int64_t)(k0 + (ch >> 4)) * row_stride + 8 * (ch & 15));
            st.v[i] = ld_frag(vglob + (int64_t)(ch >> 2) * p.t_pad + k0 + 8 * (ch & 3));
        }
    };
    auto stash = [&](int buf, const Stage &st) {
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int ch = tid + i * nthr;
            *reinterpret_cast<bf16x8 *>(&s_k[buf][(ch >> 4) * ATT_KROW + 8 * (ch & 15)]) = st.k[i];
            *reinterpret_cast<bf16x8 *>(&s_v[buf][(ch >> 2) * ATT_VROW + 8 * (ch & 3)]) = st.v[i];
        }
    };
    Stage st;
    fetch(0, st);
    stash(0, st);
    __syncthreads();
    // empty the compiler's vmcnt scoreboard: otherwise the loop keeps conservative waits on the Q fragment
    // loads above in every iteration and drains the prefetch issued at the top of each tile
    __builtin_amdgcn_s_waitcnt(0x0F70);

    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * 32, buf = kt & 1;
        if (kt + 1 < n_kt) fetch(kt + 1, st);  // in flight during this tile's MFMAs and softmax
        // all 8 K fragments first, then the MFMA chain: LDS latency is paid once, not per MFMA
        bf16x8 fr[8];
        {
            const u16 *kp = &s_k[buf][c * ATT_KROW + 8 * h];  // A[row = key c][k = 8h + j]
#pragma unroll
            for (int s = 0; s < 8; ++s) fr[s] = *reinterpret_cast<const bf16x8 *>(kp + 16 * s);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sacc = zero;
#pragma unroll
        for (int s = 0; s < 8; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], qf[s], sacc, 0, 0, 0);
        // V^T fragments into the same registers while the softmax runs:
        // A operand V^T[d = 32 dt + c][8 keys of (s2, h)], contiguous in the staged (PV-fragment) order
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                fr[2 * dt + s2] = *reinterpret_cast<const bf16x8 *>(&s_v[buf][(32 * dt + c) * ATT_VROW + 8 * h + 16 * s2]);
        __builtin_amdgcn_sched_barrier(0);
        // lane: query row q0 + c; register i: key k0 + (i&3) + 8*(i>>2) + 4h
        float sv[16];
        float mloc = -INFINITY;
        const bool diag = (kt == n_kt - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            float v = sacc[i] * p.scale_log2;
            if (diag && key > q0 + c) v = -INFINITY;
            sv[i] = v;
            mloc = fmaxf(mloc, v);
        }
        {  // combine the two half-waves (keys 4h..): v_permlane32_swap instead of an LDS-crossbar shuffle
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
            mloc = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        }
        const float mnew = fmaxf(m, mloc);  // finite: key k0 (<= q0 + c) is never masked
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sv[i] = __builtin_amdgcn_exp2f(sv[i] - mnew);
            lsum += sv[i];
        }
        {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
            lsum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        l = l * alpha + lsum;
        if (__any(mnew != m)) {  // wave-uniform: once the running maxima have settled no rescale is needed
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
        }
        m = mnew;
        // P^T fragments (B operand of k-step s2): element j = register 8*s2 + j
        bf16x8 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pf[s2][jj] = (short)f2bf(sv[8 * s2 + jj]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)  // 4 independent accumulator chains
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2 * dt + s2], pf[s2], oacc[dt], 0, 0, 0);
        if (kt + 1 < n_kt) stash(buf ^ 1, st);  // that buffer was last read in tile kt-1, before the previous barrier
        __syncthreads();
    }
    // O[q0 + c][32 dt + (i&3) + 8 (i>>2) + 4h] = oacc[dt][i] / l : 4 consecutive d per register quad
    // O[q0 + c][32 dt + (i&3) + 8 (i>>2) + 4h] = oacc[dt][i] / l.  Stored straight from these registers a wave
    // instruction would write 16 bytes into each of 32 rows (partial lines: 40 % of the kernel's time at
    // 256-token chunks); instead the wave transposes its 32 x 128 tile through LDS (the K staging buffer is free
    // after the last barrier) and writes whole 256-byte rows, 16 bytes per lane.
    {
        u16 *ot = s_pool + wave * (32 * ATT_KROW);
        const float inv = 1.f / l;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 w;
                w.x = (uint32_t)f2bf(oacc[dt][4 * g4] * inv) | ((uint32_t)f2bf(oacc[dt][4 * g4 + 1] * inv) << 16);
                w.y = (uint32_t)f2bf(oacc[dt][4 * g4 + 2] * inv) | ((uint32_t)f2bf(oacc[dt][4 * g4 + 3] * inv) << 16);
                *reinterpret_cast<uint2 *>(ot + c * ATT_KROW + 32 * dt + 8 * g4 + 4 * h) = w;
            }
        // same wave, LDS operations complete in order: no barrier between these writes and the reads below
        u16 *obase = p.out + (int64_t)(s_begin + q0) * ((int64_t)p.hq * CRAG_HEAD_DIM) + (int64_t)head * CRAG_HEAD_DIM;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (lane >> 4) + 4 * it, chunk = lane & 15;
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(ot + row * ATT_KROW + 8 * chunk);
            if (q0 + row < len)
                *reinterpret_cast<bf16x8 *>(obase + (int64_t)row * ((int64_t)p.hq * CRAG_HEAD_DIM) + 8 * chunk) = v;
        }
    }
}

// ---- 64 keys per iteration (GROUP >= 4): two 32-key tiles share one softmax update, one barrier and one
// staging round, and their QK chains interleave.  A q block with an odd number of key tiles ends with a
// half pair (HALF): only its first tile exists (its second would lie entirely above the diagonal). ----
constexpr int ATT_VROW2 = 72;  // u16 per staged V^T row of a pair (64 + 8 pad: conflict-free b128 reads)

template <int GROUP>
__global__ __launch_bounds__(64 * GROUP) __attribute__((amdgpu_waves_per_eu(2, 8)))
void attention_pair_kernel(AttnParams p) {
    static_assert(GROUP >= 4, "staging is sized for at least 256 threads");
    // one pool: K buffers, then V^T buffers; the epilogue reuses its start for the per-wave output tiles
    constexpr int K_BUF = 64 * ATT_KROW, V_BUF = CRAG_HEAD_DIM * ATT_VROW2;
    static_assert(GROUP * 32 * ATT_KROW <= 2 * (K_BUF + V_BUF), "output tiles must fit the staging pool");
    __shared__ __attribute__((aligned(16))) u16 s_pool[2 * (K_BUF + V_BUF)];
    u16(*s_k)[K_BUF] = reinterpret_cast<u16(*)[K_BUF]>(s_pool);
    u16(*s_v)[V_BUF] = reinterpret_cast<u16(*)[V_BUF]>(s_pool + 2 * K_BUF);
    constexpr int nthr = 64 * GROUP;
    constexpr int per = 1024 / nthr;  // 16-byte chunks of K and of V^T per thread and pair
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // kv head on the fast grid axis: workgroups go to the 8 XCDs round-robin by linear id, so with 8 kv heads
    // every XCD serves ONE kv head and the q blocks that re-read a sequence's K/V tiles share that XCD's L2
    const int kvh = blockIdx.x;
    const int head = kvh * GROUP + wave;
    const int seq = p.blk_seq[blockIdx.y];
    const int q0 = p.blk_q0[blockIdx.y];
    const int s_begin = p.cu[seq];
    const int len = p.cu[seq + 1] - s_begin;
    const int64_t pad_base = p.cu_pad[seq];
    const int c = lane & 31, h = lane >> 5;
    const int64_t row_stride = (int64_t)(p.hq + 2 * p.hkv) * CRAG_HEAD_DIM;

    bf16x8 qf[8];
    {
        const u16 *qp = p.qkv + (int64_t)(s_begin + q0 + c) * row_stride + (int64_t)head * CRAG_HEAD_DIM + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = ld_frag(qp + 16 * s);
    }
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 oacc[4] = {zero, zero, zero, zero};
    float m = -INFINITY, l = 0.f;
    const int n_kt = q0 / 32 + 1;          // 32-key tiles up to and including the diagonal one
    const int n_pairs = (n_kt + 1) >> 1;
    const bool last_half = (n_kt & 1) != 0;
    const u16 *kglob = p.qkv + (int64_t)s_begin * row_stride + (int64_t)(p.hq + kvh) * CRAG_HEAD_DIM;
    const u16 *vglob = p.vt + (int64_t)kvh * CRAG_HEAD_DIM * p.t_pad + pad_base;

    struct Stage {
        bf16x8 k[per], v[per];
    };
    // K pair: 64 rows x 16 chunks; V^T pair: 128 rows x 8 chunks.  A half pair moves only the first tile.
    // per-thread bases computed once; chunk i and pair pr only add wave-uniform offsets
    const u16 *kthr = kglob + (int64_t)(tid >> 4) * row_stride + 8 * (tid & 15);
    const u16 *vthr = vglob + (int64_t)(tid >> 3) * p.t_pad + 8 * (tid & 7);
    const int64_t kstep = (int64_t)(nthr >> 4) * row_stride, vstep = (int64_t)(nthr >> 3) * p.t_pad;
    auto fetch = [&](int pr, bool half, Stage &st) {
        const u16 *kp = kthr + (int64_t)(pr * 64) * row_stride;
        const u16 *vp = vthr + pr * 64;
#pragma unroll
        for (int i = 0; i < per; ++i) {
            // K rows (tid >> 4) + i * nthr / 16: the second tile's rows are i >= per / 2; V^T columns 8 * (tid & 7)
            if (!(half && i >= per / 2)) st.k[i] = ld_frag(kp + i * kstep);
            if (!(half && (tid & 7) >= 4)) st.v[i] = ld_frag(vp + i * vstep);
        }
    };
    auto stash = [&](int buf, const Stage &st) {
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int ch = tid + i * nthr;
            *reinterpret_cast<bf16x8 *>(&s_k[buf][(ch >> 4) * ATT_KROW + 8 * (ch & 15)]) = st.k[i];
            *reinterpret_cast<bf16x8 *>(&s_v[buf][(ch >> 3) * ATT_VROW2 + 8 * (ch & 7)]) = st.v[i];
        }
    };
    Stage st;
#pragma unroll
    for (int i = 0; i < per; ++i) st.k[i] = st.v[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    fetch(0, n_pairs == 1 && last_half, st);
    stash(0, st);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // clean vmcnt scoreboard at the loop head (see attention_kernel)

    // one pair of key tiles; HALF: only the first tile; DIAG: the pair's last existing tile is the diagonal one
    auto pair = [&](int pr, auto HALF_, auto DIAG_) {
        constexpr bool HALF = decltype(HALF_)::value, DIAG = decltype(DIAG_)::value;
        const int k0 = pr * 64, buf = pr & 1;
        if (pr + 1 < n_pairs) fetch(pr + 1, (pr + 2 == n_pairs) && last_half, st);
        f32x16 sa = zero, sb = zero;
        {
            const u16 *kp = &s_k[buf][c * ATT_KROW + 8 * h];
#pragma unroll
            for (int half4 = 0; half4 < 2; ++half4) {  // 4 k-steps of both tiles per round: 8 fragments in flight
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    fa[s] = *reinterpret_cast<const bf16x8 *>(kp + 16 * (4 * half4 + s));
                    if constexpr (!HALF) fb[s] = *reinterpret_cast<const bf16x8 *>(kp + 32 * ATT_KROW + 16 * (4 * half4 + s));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], qf[4 * half4 + s], sa, 0, 0, 0);
                    if constexpr (!HALF) sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s], qf[4 * half4 + s], sb, 0, 0, 0);
                }
            }
        }
        // V^T fragments of the first tile are requested now and arrive during the softmax
        bf16x8 fv[8];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                fv[2 * dt + s2] = *reinterpret_cast<const bf16x8 *>(&s_v[buf][(32 * dt + c) * ATT_VROW2 + 8 * h + 16 * s2]);
        __builtin_amdgcn_sched_barrier(0);
        // lane: query row q0 + c; register i of tile t: key k0 + 32 t + (i&3) + 8*(i>>2) + 4h
        // the softmax scale is positive, so the row maximum is taken on the raw scores and the scale is folded
        // into the exponent's fma: p = exp2(s * scale - m), m = scale * max(s)
        float mloc = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (DIAG && HALF && key > q0 + c) sa[i] = -INFINITY;
            mloc = fmaxf(mloc, sa[i]);
            if constexpr (!HALF) {
                if (DIAG && key + 32 > q0 + c) sb[i] = -INFINITY;
                mloc = fmaxf(mloc, sb[i]);
            }
        }
        {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
            mloc = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        }
        const float mnew = fmaxf(m, mloc * p.scale_log2);  // finite: key k0 (<= q0 + c) is never masked
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sa[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[i], p.scale_log2, -mnew));
            lsum += sa[i];
            if constexpr (!HALF) {
                sb[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sb[i], p.scale_log2, -mnew));
                lsum += sb[i];
            }
        }
        {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
            lsum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        l = l * alpha + lsum;
        if (__any(mnew != m)) {  // wave-uniform: once the running maxima have settled no rescale is needed
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
        }
        m = mnew;
        bf16x8 pa[2], pb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                pa[s2][jj] = (short)f2bf(sa[8 * s2 + jj]);
                if constexpr (!HALF) pb[s2][jj] = (short)f2bf(sb[8 * s2 + jj]);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)  // 4 independent accumulator chains
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[2 * dt + s2], pa[s2], oacc[dt], 0, 0, 0);
        if constexpr (!HALF) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    fv[2 * dt + s2] = *reinterpret_cast<const bf16x8 *>(&s_v[buf][(32 * dt + c) * ATT_VROW2 + 8 * h + 32 + 16 * s2]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[2 * dt + s2], pb[s2], oacc[dt], 0, 0, 0);
        }
        if (pr + 1 < n_pairs) stash(buf ^ 1, st);  // that buffer was last read one barrier ago
        __syncthreads();
    };

    for (int pr = 0; pr + 1 < n_pairs; ++pr) pair(pr, std::false_type{}, std::false_type{});
    if (last_half) pair(n_pairs - 1, std::true_type{}, std::true_type{});
    else pair(n_pairs - 1, std::false_type{}, std::true_type{});

    // O[q0 + c][32 dt + (i&3) + 8 (i>>2) + 4h] = oacc[dt][i] / l.  Stored straight from these registers a wave
    // instruction would write 16 bytes into each of 32 rows (partial lines: 40 % of the kernel's time at
    // 256-token chunks); instead the wave transposes its 32 x 128 tile through LDS (the K staging buffer is free
    // after the last barrier) and writes whole 256-byte rows, 16 bytes per lane.
    {
        u16 *ot = s_pool + wave * (32 * ATT_KROW);
        const float inv = 1.f / l;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 w;
                w.x = (uint32_t)f2bf(oacc[dt][4 * g4] * inv) | ((uint32_t)f2bf(oacc[dt][4 * g4 + 1] * inv) << 16);
                w.y = (uint32_t)f2bf(oacc[dt][4 * g4 + 2] * inv) | ((uint32_t)f2bf(oacc[dt][4 * g4 + 3] * inv) << 16);
                *reinterpret_cast<uint2 *>(ot + c * ATT_KROW + 32 * dt + 8 * g4 + 4 * h) = w;
            }
        // same wave, LDS operations complete in order: no barrier between these writes and the reads below
        u16 *obase = p.out + (int64_t)(s_begin + q0) * ((int64_t)p.hq * CRAG_HEAD_DIM) + (int64_t)head * CRAG_HEAD_DIM;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (lane >> 4) + 4 * it, chunk = lane & 15;
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(ot + row * ATT_KROW + 8 * chunk);
            if (q0 + row < len)
                *reinterpret_cast<bf16x8 *>(obase + (int64_t)row * ((int64_t)p.hq * CRAG_HEAD_DIM) + 8 * chunk) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SwiGLU
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_kernel(const u16 *gu, u16 *out, int64_t rows, int inter) {
    // one row per block trip, 16-byte chunks across the threads: no index division, all loads of a trip in flight
    const int chunks = inter >> 3;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const Pack8 *gp = reinterpret_cast<const Pack8 *>(gu + r * 2 * inter);
        const Pack8 *up = reinterpret_cast<const Pack8 *>(gu + r * 2 * inter + inter);
        Pack8 *op = reinterpret_cast<Pack8 *>(out + r * inter);
        for (int c0 = threadIdx.x; c0 < chunks; c0 += 4 * 256) {
            Pack8 g[4], u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + q * 256;
                if (c < chunks) {
                    g[q] = gp[c];
                    u[q] = up[c];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + q * 256;
                if (c < chunks) {
                    Pack8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = bf2f(g[q].v[e]);
                        const float act = bf2f(f2bf(x / (1.f + __expf(-x))));  // silu in fp32, rounded to bf16 like the model
                        o.v[e] = f2bf(act * bf2f(u[q].v[e]));
                    }
                    op[c] = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pooling + final norm + slice + L2 normalise (one block per sequence)
// ---------------------------------------------------------------------------------------------
// `delta` (nullable, mode 0): the last sub-block's output, added to the residual stream for the pooled rows only
// (bf16 add, as the model's `hidden + mlp(...)`), instead of a pass over every token just to pool one per sequence.
// `rows` (nullable, mode 0): the pooled row of sequence b is rows[b] instead of the sequence's last row (a graph replay
// pools at the real last token of every padded sequence: data, not shape).
__global__ __launch_bounds__(256) void pool_normalize_kernel(const u16 *hs, const u16 *delta, const u16 *w, const int32_t *cu,
                                                             const int64_t *rows, float *out, int hidden, int out_dim,
                                                             int mode, float eps) {
    __shared__ float sh[4];
    __shared__ float row[8192];
    const int b = blockIdx.x;
    const int t0 = rows ? (int)rows[b] : cu[b], t1 = rows ? t0 + 1 : cu[b + 1];
    if (t1 <= t0) {
        for (int i = threadIdx.x; i < out_dim; i += blockDim.x) out[(int64_t)b * out_dim + i] = 0.f;
        return;
    }
    if (mode == 0) {  // last token of the residual stream, final RMSNorm applied here
        const u16 *x = hs + (int64_t)(t1 - 1) * hidden;
        const u16 *dl = delta ? delta + (int64_t)(t1 - 1) * hidden : nullptr;
        float ss = 0.f;
        for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
            const float v = dl ? bf2f(f2bf(bf2f(x[i]) + bf2f(dl[i]))) : bf2f(x[i]);
            row[i] = v;
            ss += v * v;
        }
        ss = block_sum(ss, sh);
        const float rstd = rsqrtf(ss / (float)hidden + eps);
        for (int i = threadIdx.x; i < hidden; i += blockDim.x)
            row[i] = bf2f(f2bf(bf2f(w[i]) * bf2f(f2bf(row[i] * rstd))));
    } else {  // mean over the (already normed) tokens
        const float invn = 1.f / (float)(t1 - t0);
        for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
            float acc = 0.f;
            for (int t = t0; t < t1; ++t) acc += bf2f(hs[(int64_t)t * hidden + i]);
            row[i] = acc * invn;
        }
    }
    __syncthreads();
    float ss = 0.f;
    for (int i = threadIdx.x; i < out_dim; i += blockDim.x) ss += row[i] * row[i];
    ss = block_sum(ss, sh);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);  // RUNBOOK:510-513 max(norm, 1e-12)
    for (int i = threadIdx.x; i < out_dim; i += blockDim.x) out[(int64_t)b * out_dim + i] = row[i] * inv;
}


// ---------------------------------------------------------------------------------------------
// Skinny GEMM: out[M, N] = x[M, K] @ W[N, K]^T for M <= 32 tokens -- the encoder at the reference's own operating
// point, ONE query per /retrieve request (retrieve.py:427).  At that size a linear layer is a WEIGHT STREAM (202 MB of
// bf16 weights per decoder layer against 16-32 rows of activations): HBM bound, 25 us per layer at 8 TB/s, where the
// library's GEMM kernels for 16 rows measured ~1 TB/s (profiles/r03_small_encode_kernel_stats.csv).
//   * weights are stored a second time in MFMA A-fragment order (Qwen3Encoder._small_weights):
//     wsw[n_tile of 16 rows][k-step of 32][lane = 16 (k/8) + row][8 bf16], so a wave-instruction reads 1 KiB of
//     contiguous HBM and the loaded registers ARE the A operand of v_mfma_f32_16x16x32_bf16 (rows = output features);
//   * the activations are the B operand (columns = tokens), each wave keeps the fragments of ITS K range in registers
//     for the whole kernel (x is 80-300 KB, L2 resident); the K dimension is split over the WAVES waves of a workgroup,
//     the partial 16x16 tiles are summed through LDS in wave order (deterministic);
//   * a workgroup owns NT n-tiles; its waves keep 12 weight loads (12 KiB per wave) in flight;
//   * EPI = 1: the rows of an n-tile are 8 gate rows followed by the 8 up rows of the same features (weights
//     interleaved that way), and the epilogue writes silu(gate) * up with the model's bf16 roundings (gate and up
//     rounded to bf16 as a linear layer's output would be, silu in fp32 rounded to bf16, product rounded to bf16).
// ---------------------------------------------------------------------------------------------
struct SkinnyParams {
    const u16 *x;    // [16 * MG, K] bf16, rows >= m_rows are padding (never stored)
    const u16 *wsw;  // fragment-ordered weights
    u16 *out;        // [m_rows, ld_out] bf16
    int m_rows, n, k, ld_out;
};

template <int MG, int NT, int KS, int WAVES, int EPI>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(SkinnyParams p) {
    __shared__ f32x4_t red[WAVES][NT][MG][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ksteps = p.k >> 5;
    const int kw0 = w * KS;  // first k-step of this wave
    // activations: B[k = 8 (lane >> 4) + j][col = lane & 15] = x[token 16 mg + (lane & 15)][32 (kw0 + s) + 8 (lane >> 4) + j]
    bf16x8 xb[MG][KS];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        const u16 *xr = p.x + (size_t)(16 * mg + (lane & 15)) * p.k + (size_t)kw0 * 32 + 8 * (lane >> 4);
#pragma unroll
        for (int s = 0; s < KS; ++s) xb[mg][s] = *reinterpret_cast<const bf16x8 *>(xr + 32 * s);
    }
    constexpr int TOT = NT * KS, D = TOT < 12 ? TOT : 12;
    const u16 *wbase = p.wsw + ((size_t)blockIdx.x * NT * ksteps + (size_t)kw0) * 512 + lane * 8;
    auto wptr = [&](int idx) -> const bf16x8 * {
        const int nt = idx / KS, s = idx % KS;
        return reinterpret_cast<const bf16x8 *>(wbase + ((size_t)nt * ksteps + s) * 512);
    };
    bf16x8 wr[D];
#pragma unroll
    for (int i = 0; i < D; ++i) wr[i] = __builtin_nontemporal_load(wptr(i));
    f32x4_t acc[NT][MG];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) acc[nt][mg] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TOT; ++i) {
        const int nt = i / KS, s = i % KS;
        const bf16x8 wv = wr[i % D];
        if (i + D < TOT) wr[i % D] = __builtin_nontemporal_load(wptr(i + D));
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
            acc[nt][mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, xb[mg][s], acc[nt][mg], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) red[w][nt][mg][lane] = acc[nt][mg];
    __syncthreads();
    // D[row = 4 (lane >> 4) + reg][col = lane & 15]: a (tile, token group, lane) unit = 4 consecutive features of one token
    constexpr int UNITS = NT * MG * 64;
    for (int u = threadIdx.x; u < UNITS; u += WAVES * 64) {
        const int l = u & 63, mg = (u >> 6) % MG, nt = (u >> 6) / MG;
        const int token = 16 * mg + (l & 15);
        if (EPI == 1 && l >= 32) continue;  // the up half is consumed by the thread of its gate half
        f32x4_t sum = red[0][nt][mg][l];
#pragma unroll
        for (int ww = 1; ww < WAVES; ++ww) sum += red[ww][nt][mg][l];
        if (token >= p.m_rows) continue;
        const int tile = (int)blockIdx.x * NT + nt;
        if (EPI == 0) {
            u16 o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(sum[r]);
            *reinterpret_cast<uint2 *>(p.out + (size_t)token * p.ld_out + 16 * tile + 4 * (l >> 4)) =
                make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        } else {
            f32x4_t up = red[0][nt][mg][l + 32];
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) up += red[ww][nt][mg][l + 32];
            u16 o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = bf2f(f2bf(sum[r]));
                const float act = bf2f(f2bf(g / (1.f + __expf(-g))));
                o[r] = f2bf(act * bf2f(f2bf(up[r])));
            }
            *reinterpret_cast<uint2 *>(p.out + (size_t)token * p.ld_out + 8 * tile + 4 * (l >> 4)) =
                make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        }
    }
}

}  // namespace

extern "C" {

int crag_enc_embed_gather(const int32_t *ids, const uint16_t *table, uint16_t *out, int64_t n_tokens, int hidden,
                          int64_t vocab, void *stream) {
    if (!ids || !table || !out) return efail("embed_gather: NULL pointer");
    if (hidden <= 0 || (hidden & 7) || vocab <= 0 || n_tokens < 0) return efail("embed_gather: bad sizes");
    if (n_tokens == 0) return 0;
    const unsigned grid = (unsigned)(n_tokens < 65535 * 16 ? n_tokens : 65535 * 16);
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, table, out, n_tokens,
                       hidden, vocab);
    return hip_ok("embed_gather");
}

int crag_enc_rmsnorm(const uint16_t *x, const uint16_t *residual_in, const uint16_t *weight, uint16_t *out,
                     uint16_t *residual_out, int64_t rows, int hidden, float eps, void *stream) {
    if (!x || !weight || !out) return efail("rmsnorm: NULL pointer");
    if (hidden <= 0 || (hidden & 7) || hidden > 8 * NORM_THREADS * NORM_MAX_CHUNKS)
        return efail("rmsnorm: hidden must be a multiple of 8 and <= %d (got %d)", 8 * NORM_THREADS * NORM_MAX_CHUNKS, hidden);
    if (rows <= 0) return 0;
    const unsigned grid = (unsigned)(rows < 1048576 ? rows : 1048576);
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(grid), dim3(NORM_THREADS), 0, (hipStream_t)stream, x, residual_in, weight,
                       out, residual_out, rows, hidden, eps);
    return hip_ok("rmsnorm");
}

int crag_enc_qk_norm_rope(uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w, const float *cos_sin,
                          const int32_t *positions, int64_t n_tokens, int hq, int hkv, float eps, void *stream) {
    if (!qkv || !q_norm_w || !k_norm_w || !cos_sin || !positions) return efail("qk_norm_rope: NULL pointer");
    if (hq <= 0 || hkv <= 0) return efail("qk_norm_rope: bad head counts");
    if (n_tokens <= 0) return 0;
    const unsigned grid = (unsigned)(n_tokens < 65536 * 16 ? n_tokens : 65536 * 16);
    hipLaunchKernelGGL(qk_norm_rope_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, qkv,
                       q_norm_w, k_norm_w, cos_sin, positions, n_tokens, hq, hkv, eps);
    return hip_ok("qk_norm_rope");
}

int crag_enc_v_transpose(const uint16_t *qkv, uint16_t *vt, const int32_t *tok_of_pad, int64_t t_pad, int hq, int hkv,
                         void *stream) {
    if (!qkv || !vt || !tok_of_pad) return efail("v_transpose: NULL pointer");
    if (t_pad <= 0 || (t_pad & 31)) return efail("v_transpose: t_pad must be a positive multiple of 32");
    hipLaunchKernelGGL(v_transpose_kernel, dim3((unsigned)(t_pad / 32), (unsigned)hkv), dim3(256), 0,
                       (hipStream_t)stream, qkv, vt, tok_of_pad, t_pad, hq, hkv);
    return hip_ok("v_transpose");
}

int crag_enc_qk_rope_vt(uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w, const float *cos_sin,
                        const int32_t *positions, int64_t n_tokens, int hq, int hkv, float eps, uint16_t *vt,
                        const int32_t *tok_of_pad, int64_t t_pad, void *stream) {
    if (!qkv || !q_norm_w || !k_norm_w || !cos_sin || !positions || !vt || !tok_of_pad) return efail("qk_rope_vt: NULL pointer");
    if (hq <= 0 || hkv <= 0) return efail("qk_rope_vt: bad head counts");
    if (t_pad <= 0 || (t_pad & 31)) return efail("qk_rope_vt: t_pad must be a positive multiple of 32");
    if (n_tokens <= 0) return 0;
    const int rope_blocks = (int)(n_tokens < 65536 ? n_tokens : 65536);
    const unsigned grid = (unsigned)rope_blocks + (unsigned)(t_pad / 32) * (unsigned)hkv;
    hipLaunchKernelGGL(qk_rope_vt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, qkv, q_norm_w, k_norm_w,
                       cos_sin, positions, n_tokens, hq, hkv, eps, vt, tok_of_pad, t_pad, rope_blocks);
    return hip_ok("qk_rope_vt");
}

int crag_enc_attention(const uint16_t *qkv, const uint16_t *vt, uint16_t *out, const int32_t *cu_seqlens,
                       const int32_t *cu_pad, const int32_t *blk_seq, const int32_t *blk_q0, int n_blocks,
                       int64_t t_pad, int hq, int hkv, float scale, void *stream) {
    if (!qkv || !vt || !out || !cu_seqlens || !cu_pad || !blk_seq || !blk_q0) return efail("attention: NULL pointer");
    if (hkv <= 0 || hq % hkv != 0 || (hq / hkv != 1 && hq / hkv != 2 && hq / hkv != 4 && hq / hkv != 8))
        return efail("attention: hq/hkv must be 1, 2, 4 or 8");
    if (n_blocks <= 0) return 0;
    AttnParams p;
    p.qkv = qkv;
    p.vt = vt;
    p.out = out;
    p.cu = cu_seqlens;
    p.cu_pad = cu_pad;
    p.blk_seq = blk_seq;
    p.blk_q0 = blk_q0;
    p.t_pad = t_pad;
    p.hq = hq;
    p.hkv = hkv;
    p.scale_log2 = scale * 1.4426950408889634f;
    if (n_blocks > 65535 * 64) return efail("attention: too many q blocks (%d)", n_blocks);
    const dim3 grid((unsigned)hkv, (unsigned)n_blocks);
    switch (hq / hkv) {
        case 1: hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(64), 0, (hipStream_t)stream, p); break;
        case 2: hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(128), 0, (hipStream_t)stream, p); break;
        case 4:
            if (getenv("CRAG_ATTN_SINGLE")) hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
            else hipLaunchKernelGGL(attention_pair_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
            break;
        default: hipLaunchKernelGGL(attention_pair_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, p); break;
    }
    return hip_ok("attention");
}

int crag_enc_swiglu(const uint16_t *gate_up, uint16_t *out, int64_t rows, int inter, void *stream) {
    if (!gate_up || !out) return efail("swiglu: NULL pointer");
    if (inter <= 0 || (inter & 7)) return efail("swiglu: inter must be a positive multiple of 8");
    if (rows <= 0) return 0;
    const unsigned grid = (unsigned)(rows < 65536 * 16 ? rows : 65536 * 16);
    hipLaunchKernelGGL(swiglu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gate_up, out, rows, inter);
    return hip_ok("swiglu");
}

int crag_enc_pool_normalize(const uint16_t *hidden_states, const uint16_t *final_norm_w, const int32_t *cu_seqlens,
                            float *out, int n_seqs, int hidden, int out_dim, int mode, float eps, void *stream) {
    return crag_enc_pool_normalize_add(hidden_states, nullptr, final_norm_w, cu_seqlens, out, n_seqs, hidden, out_dim,
                                       mode, eps, stream);
}

int crag_enc_pool_normalize_add(const uint16_t *hidden_states, const uint16_t *delta, const uint16_t *final_norm_w,
                                const int32_t *cu_seqlens, float *out, int n_seqs, int hidden, int out_dim, int mode,
                                float eps, void *stream) {
    if (!hidden_states || !cu_seqlens || !out) return efail("pool_normalize: NULL pointer");
    if (mode == 0 && !final_norm_w) return efail("pool_normalize: last-token mode needs the final norm weight");
    if (delta && mode != 0) return efail("pool_normalize: a delta is only added in last-token mode");
    if (hidden <= 0 || hidden > 8192 || out_dim <= 0 || out_dim > hidden) return efail("pool_normalize: bad sizes");
    if (n_seqs <= 0) return 0;
    hipLaunchKernelGGL(pool_normalize_kernel, dim3((unsigned)n_seqs), dim3(256), 0, (hipStream_t)stream, hidden_states,
                       delta, final_norm_w, cu_seqlens, (const int64_t *)nullptr, out, hidden, out_dim, mode, eps);
    return hip_ok("pool_normalize");
}

int crag_enc_pool_normalize_rows(const uint16_t *hidden_states, const uint16_t *delta, const uint16_t *final_norm_w,
                                 const int64_t *rows, float *out, int n_seqs, int hidden, int out_dim, float eps,
                                 void *stream) {
    if (!hidden_states || !final_norm_w || !rows || !out) return efail("pool_normalize_rows: NULL pointer");
    if (hidden <= 0 || hidden > 8192 || out_dim <= 0 || out_dim > hidden) return efail("pool_normalize_rows: bad sizes");
    if (n_seqs <= 0) return 0;
    hipLaunchKernelGGL(pool_normalize_kernel, dim3((unsigned)n_seqs), dim3(256), 0, (hipStream_t)stream, hidden_states,
                       delta, final_norm_w, (const int32_t *)nullptr, rows, out, hidden, out_dim, 0, eps);
    return hip_ok("pool_normalize_rows");
}

int crag_enc_skinny_gemm(const uint16_t *x, const uint16_t *wsw, uint16_t *out, int m_rows, int m_pad, int n, int k,
                         int epilogue, void *stream) {
    if (!x || !wsw || !out) return efail("skinny_gemm: NULL pointer");
    if (m_rows <= 0 || m_rows > m_pad || (m_pad != 16 && m_pad != 32)) return efail("skinny_gemm: m_pad must be 16 or 32 and m_rows <= m_pad");
    if (epilogue != 0 && epilogue != 1) return efail("skinny_gemm: epilogue must be 0 or 1");
    if (n <= 0 || (n & 15)) return efail("skinny_gemm: n must be a positive multiple of 16");
    SkinnyParams p;
    p.x = x;
    p.wsw = wsw;
    p.out = out;
    p.m_rows = m_rows;
    p.n = n;
    p.k = k;
    p.ld_out = epilogue ? n / 2 : n;
    const int tiles = n / 16;
    hipStream_t st = (hipStream_t)stream;
#define CRAG_SKINNY(MG_, NT_, KS_, WAVES_, EPI_)                                                              \
    hipLaunchKernelGGL((skinny_gemm_kernel<MG_, NT_, KS_, WAVES_, EPI_>), dim3((unsigned)(tiles / NT_)),      \
                       dim3(WAVES_ * 64), 0, st, p)
    const int mg = m_pad / 16;
    // n-tiles per workgroup (NT): a workgroup reads the activations of its K range once per NT tiles, but tiles / NT
    // workgroups must still cover the chip -- measured per shape (scripts/probes/skinny_bench.py): two tiles for the
    // wide projections (qkv 10.1 vs 10.7 us, gate|up 22.4 vs 23.1 with four), one for o and down (160 n-tiles: with
    // two tiles per workgroup only 80 CUs work, 12.3 vs 8.4 us and 24.1 vs 16.7)
    if (k == 2560 && epilogue == 0) {
        if (tiles & 1) return efail("skinny_gemm: n / 16 must be even for k = 2560");
        if (mg == 1) CRAG_SKINNY(1, 2, 10, 8, 0); else CRAG_SKINNY(2, 2, 10, 8, 0);
    } else if (k == 2560 && epilogue == 1) {
        if (tiles & 1) return efail("skinny_gemm: the SwiGLU form needs an even number of n-tiles");
        if (mg == 1) CRAG_SKINNY(1, 2, 10, 8, 1); else CRAG_SKINNY(2, 2, 10, 8, 1);
    } else if (k == 4096 && epilogue == 0) {
        if (mg == 1) CRAG_SKINNY(1, 1, 16, 8, 0); else CRAG_SKINNY(2, 1, 16, 8, 0);
    } else if (k == 9728 && epilogue == 0) {
        if (mg == 1) CRAG_SKINNY(1, 1, 38, 8, 0); else CRAG_SKINNY(2, 1, 19, 16, 0);
    } else {
        return efail("skinny_gemm: unsupported shape k=%d epilogue=%d (built for the Qwen3-Embedding-4B widths: k = 2560 / 4096 / 9728)", k, epilogue);
    }
#undef CRAG_SKINNY
    return hip_ok("skinny_gemm");
}

}  // extern "C"
