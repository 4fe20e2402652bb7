// crag_encoder_wide.hip — the linear layers of the Qwen3-Embedding decoder at 64 / 128 token rows: what the gateway's
// dynamic batcher hands the model (/root/reference/P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:304,331-334:
// max_batch_size 8, preferred_batch_size [1, 2, 4, 8] -- 3 to 8 short queries, or one query of 33 to 128 tokens).
// C ABI: include/crag_encoder.h (crag_enc_wide_gemm, crag_enc_wide_reduce).
//
// At this size a layer is still a WEIGHT STREAM (202 MB of bf16 weights against <= 128 rows of activations: 25 us at
// 8 TB/s, 16 GFLOP = 6 us of MFMA time), but the activations no longer fit a wave's registers (crag_encoder_small.hip
// keeps a wave's K slice of 16 / 32 rows there), and the library's small-M GEMM tiles stream the weights at 1-3 TB/s.
//
// WHERE IT STANDS (round 4, measured as hipGraph replays over rotating weight copies, scripts/probes/wide_gemm_bench.py,
// profiles/r04_wide_gemm.txt; library = hipBLASLt through torch.matmul, + crag_enc_swiglu for gate|up):
//   128 rows: qkv 20.1 us (library 15.4), o 17.7 (18.8), gate|up + SwiGLU 29.0 in ONE launch (37.2), down 24.1 (39.3)
//    64 rows: qkv 12.8 (11.3), o 11.4 (10.9), gate|up + SwiGLU 23.5 (30.9), down 18.9 (21.7)
// The encoder therefore takes gate|up and down at 64 / 128 rows from here and leaves qkv / o to the library
// (Qwen3Encoder._wide_weights): 8 queries of 16 tokens 5.47 -> 4.25 ms, 4 queries 4.03 -> 3.51 ms
// (scripts/probes/small_batch_encode.py).  What was learned on the way (all measured, each cost a factor):
//   (1) hipcc's own load bookkeeping cannot pipeline this loop: `if (more)` around a load makes it wait vmcnt(0) right
//       behind that load (6 GB/s per workgroup); without the branch it waits vmcnt(0) at the head of every iteration
//       for loads issued in an earlier one and sinks the next chunk's loads to the bottom of this one (1.6-1.9 us per
//       chunk against 0.43 us of MFMAs).  The loads are inline asm with hand-counted waits now.
//   (2) the 16 pieces of a token row fall on the same four LDS banks unless the staging is swizzled (1.7 us of ds_write
//       per chunk).
//   (3) fixed cost per projection: 1.8 us launch + 2.4 us first loads + 5.5 us to write 12.6 MB of fp32 partial tiles
//       (qkv, splitk 4) + a 4-5 us reduce launch, against 8 us of streaming: the unsplit form writes its output itself
//       (crag_enc_wide_gemm_direct: gate|up), and what still needs split K (N = 2560: 20 tiles) pays for it.
// Next: the split-K partials summed in the prologue of the kernel that consumes them (the residual-add + RMSNorm pass
// for o / down, q/k-norm + RoPE for qkv) instead of a reduce launch; 64-row tiles (2 waves) for the N = 2560 projections.
//
//   * A workgroup = 4 waves = 128 rows of the weight (each wave its own 32 rows: the A operand of
//     v_mfma_f32_32x32x16_bf16, streamed once from HBM straight into registers, 8 fragments = 8 KiB in flight per
//     wave, stored in that order by wide_weight()); the waves share the B operand: a 128-column chunk of the
//     activations, staged through LDS in B-fragment order (full 256-byte rows from L2, conflict-free ds_read_b128),
//     double-buffered, one barrier per chunk.  Per 16-element k-step a wave loads 1 KiB of weights and reads MG KiB of
//     LDS for MG MFMAs (MG = rows / 32): bytes of activations per byte of weight = MG, the same for every projection.
//   * N / 128 is 20 (o, down) to 152 (gate|up) tiles: to occupy the chip the K range is SPLIT over `splitk` workgroups
//     per tile.  Each writes its fp32 partial tile in register order (one contiguous KiB per wave-instruction);
//     crag_enc_wide_reduce sums the splits in a fixed order, rounds to bf16 once (the rounding point of the unsplit
//     GEMM) and applies the SwiGLU epilogue where asked.  [An in-launch "last arriver reduces" is ruled out by the
//     slab size: splitk x 64 KiB per tile for ONE workgroup to read is 5-30 us of serial tail; a separate pass over the
//     whole output uses every CU for ~10 MB.]
//
// Arithmetic: bf16 operands, fp32 accumulation (MFMA), the K splits added in split order in fp32, one rounding to
// bf16; SwiGLU with crag_enc_swiglu's roundings (bf16 gate -> silu in fp32 -> bf16 -> x bf16 up -> bf16).

#include "crag_arch.h"
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/crag_encoder.h"

extern "C" void crag_set_error_(const char *msg);  // crag_api.hip

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint16_t u16;

int wfail(const char *fmt, ...) {
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    crag_set_error_(buf);
    return -1;
}

int whip_ok(const char *what) {
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

constexpr int WIDE_BK = 128;            // K granularity of the interface (chunks are 128 or 64 columns inside)
constexpr int WIDE_THREADS = 256;       // (the reduce kernel's block)

struct WideParams {
    const u16 *x;      // [32 MG, K] bf16 activations (rows beyond the real ones are padding: finite, results dropped)
    const u16 *ww;     // weights, [N / 32][K / 16][64 lanes][8]: lane l = row (l & 31), k = 8 (l >> 5) + e
    float *partial;    // [splitk][N / 32][MG][4][64][4] fp32: the accumulator registers as they are (NULL: direct output)
    u16 *out;          // direct output (splitk == 1): [m_rows, ld_out] bf16
    int K, n32, splitk;
    int m_rows, ld_out, epilogue;
    int token_major;   // partial as [splitk][32 MG tokens][N] fp32 rows (what crag_enc_rmsnorm_partials reads) instead
};

// bf16 output of one wave's 32 x 32 accumulator tile (D[row 8 b + 4 h + c][token] = acc[4 b + c]); epilogue 1: the 32
// rows are 16 gate rows then the 16 up rows of the same features: out[token][16 tile + f] = silu(gate_f) * up_f
__device__ __forceinline__ void wide_store(u16 *out, int ld_out, int n32, int token, int h, const f32x4_t (&sum)[4], int epilogue) {
    if (epilogue == 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {   // rows 8 b + 4 h + 0..3
            uint2 o;
            o.x = (uint32_t)f2bf(sum[b][0]) | ((uint32_t)f2bf(sum[b][1]) << 16);
            o.y = (uint32_t)f2bf(sum[b][2]) | ((uint32_t)f2bf(sum[b][3]) << 16);
            *reinterpret_cast<uint2 *>(out + (size_t)token * ld_out + 32 * n32 + 8 * b + 4 * h) = o;
        }
    } else {
#pragma unroll
        for (int b = 0; b < 2; ++b) {   // gate rows 8 b + 4 h + c, up rows 16 + the same
            u16 o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gt = bf2f(f2bf(sum[b][c]));
                const float act = bf2f(f2bf(gt / (1.f + __expf(-gt))));
                o[c] = f2bf(act * bf2f(f2bf(sum[b + 2][c])));
            }
            *reinterpret_cast<uint2 *>(out + (size_t)token * ld_out + 16 * n32 + 8 * b + 4 * h) =
                make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        }
    }
}

// MG token groups of 32; WV waves per workgroup (4: 128 rows of W per workgroup, 2: 64 rows -- twice the workgroups for
// the same N, at twice the activation bytes per byte of weight); WKS k-steps of 16 per LDS chunk (8 with 4 waves, 4 with 2:
// a thread stages MG * WKS / WV = 8 pieces of a 128-row chunk either way)
template <int MG, int WV, int WKS>
__global__ __launch_bounds__(64 * WV) void wide_gemm_kernel(WideParams p) {
    constexpr int WIDE_KS = WKS, WIDE_BK = 16 * WKS, WIDE_THREADS = 64 * WV;   // (shadow the 4-wave constants)
    __shared__ bf16x8 xs[2][WIDE_KS][MG][64];   // B fragments of a chunk: lane l = token (l & 31), k = 8 (l >> 5) + e
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = (int)blockIdx.x * WV + w;          // this wave's 32 weight rows
    const int split = blockIdx.y;
    const int chunks = p.K / WIDE_BK;
    const int c0 = (int)((long long)chunks * split / p.splitk), c1 = (int)((long long)chunks * (split + 1) / p.splitk);
    const bool rows_ok = n32 < p.n32;                  // (N is a multiple of 128 for every projection; kept for safety)

    // activation staging: piece q of 16 bytes = x[token][128 c + 8 (q & 15) .. + 7], token = q >> 4; a row of a chunk is
    // 256 contiguous bytes = 16 consecutive threads.  It lands at fragment (k-step ks = (q & 15) >> 1, group token >> 5),
    // lane 32 h + (token & 31) with h = q & 1 -- XOR-swizzled inside its 16-lane group by (ks | h << 3): the 16 pieces of
    // a token (one phase of a ds_write_b128) would otherwise all fall on the same four banks (their addresses differ by
    // multiples of 512 bytes): a 16-way conflict, 1.7 us of LDS time per chunk against 0.43 us of MFMAs.  The readers
    // apply the same XOR (a permutation inside each 16-lane phase: their reads stay conflict-free).
    constexpr int PPR = 2 * WIDE_KS;                              // 16-byte pieces per token row and chunk
    constexpr int PIECES = MG * 32 * PPR / WIDE_THREADS;          // per thread: 2 (32 rows), 4 (64), 6 (96) or 8 (128 rows)
    static_assert(PIECES == 2 || PIECES == 4 || PIECES == 6 || PIECES == 8, "the staging waits name their registers");
    const u16 *xsrc[PIECES];
    bf16x8 *xdst[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int q = tid + WIDE_THREADS * i, token = q / PPR, piece = q % PPR;
        xsrc[i] = p.x + (size_t)token * p.K + piece * 8;
        xdst[i] = &xs[0][piece >> 1][token >> 5][(32 * (piece & 1) + (token & 31)) ^ ((piece >> 1) | ((piece & 1) << 3))];
    }
    constexpr int BUF_STRIDE = WIDE_KS * MG * 64;          // bf16x8 elements between the two LDS buffers
    const int lsw = lane ^ ((lane >> 5) << 3);             // this lane's slot in a fragment of k-step 0 (k-step s: ^ s)
    const bf16x8 *wsrc = reinterpret_cast<const bf16x8 *>(p.ww) + ((size_t)(rows_ok ? n32 : 0) * (p.K / 16)) * 64 + lane;

    f32x16 acc[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mg][r] = 0.f;
    if (c0 < c1) {
        // TWO chunks in flight per wave, weights (16 x 1 KiB) and activations alike, each in two register sets used in
        // turn (the chunk loop is unrolled by two).  The loads are inline asm with HAND-COUNTED waits: hipcc's own
        // bookkeeping waits vmcnt(0) at the head of every loop iteration for a load issued in an earlier one ("pending
        // since unknown") and sinks the next chunk's loads to the bottom of this one -- whatever the source said, the
        // ring was one chunk deep: 1.6-1.9 us per chunk against 0.43 us of MFMAs.  (Earlier forms of the same trap:
        // `if (c + 1 < c1)` around a load makes hipcc wait vmcnt(0) right behind it.)  Every load is therefore
        // UNCONDITIONAL -- past the end a chunk re-loads the last one, an L2 hit nobody uses -- and the counts are the
        // same in every iteration.  Program order of the loads of chunk c: NX activation pieces for chunk c + 2, then
        // one weight fragment for chunk c + 2 behind each k-step's MFMAs.  Hence, in steady state and from the prologue
        // on (KS k-steps per chunk):
        //      * the weight fragment of k-step s was requested 2 chunks ago; behind it came KS - 1 - s + (NX + KS) + NX + s
        //        loads: vmcnt(2 KS - 1 + 2 NX) in front of every k-step;
        //      * the pieces staged at the end of chunk c (for chunk c + 1) were requested at the top of chunk c - 1;
        //        behind them came KS + NX + KS loads: vmcnt(2 KS + NX).
        constexpr int NX = PIECES, W_WAIT = 2 * WIDE_KS - 1 + 2 * NX, X_WAIT = 2 * WIDE_KS + NX;
        const int clast = c1 - 1;
        auto gload = [](bf16x8 &dst, const void *ptr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory"); };
        auto gload_nt = [](bf16x8 &dst, const void *ptr) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(ptr) : "memory"); };
        bf16x8 wa[WIDE_KS], wb[WIDE_KS], xa[PIECES], xb[PIECES];
        const int cb = c0 + 1 < c1 ? c0 + 1 : clast;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) gload(xa[i], xsrc[i] + (size_t)c0 * WIDE_BK);
#pragma unroll
        for (int s = 0; s < WIDE_KS; ++s) gload_nt(wa[s], wsrc + (size_t)(c0 * WIDE_KS + s) * 64);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) gload(xb[i], xsrc[i] + (size_t)cb * WIDE_BK);
#pragma unroll
        for (int s = 0; s < WIDE_KS; ++s) gload_nt(wb[s], wsrc + (size_t)(cb * WIDE_KS + s) * 64);
        if constexpr (PIECES == 8)
            asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), "+v"(xa[7]) : "n"(X_WAIT));
        else if constexpr (PIECES == 6)
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]) : "n"(X_WAIT));
        else if constexpr (PIECES == 4)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]) : "n"(X_WAIT));
        else
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xa[0]), "+v"(xa[1]) : "n"(X_WAIT));
#pragma unroll
        for (int i = 0; i < PIECES; ++i) xdst[i][0] = xa[i];
        __syncthreads();
        int buf = 0;
        // chunk c: computes from LDS buffer `buf`, requests chunk c + 2 (pieces into `xload`, fragments back into `wr`),
        // stages chunk c + 1 (`xstage`, requested during chunk c - 1) into the other buffer
        auto chunk = [&](int c, bf16x8 (&wr)[WIDE_KS], bf16x8 (&xload)[PIECES], bf16x8 (&xstage)[PIECES]) {
            const int cw = c + 2 < c1 ? c + 2 : clast;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) gload(xload[i], xsrc[i] + (size_t)cw * WIDE_BK);
            bf16x8 bq[2][MG];   // the B fragments of a k-step are read one k-step ahead of their MFMAs
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) bq[0][mg] = xs[buf][0][mg][lsw];
#pragma unroll
            for (int s = 0; s < WIDE_KS; ++s) {
                if (s + 1 < WIDE_KS) {
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg) bq[(s + 1) & 1][mg] = xs[buf][s + 1][mg][lsw ^ (s + 1)];
                }
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wr[s]) : "n"(W_WAIT));
#pragma unroll
                for (int mg = 0; mg < MG; ++mg)
                    acc[mg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[s], bq[s & 1][mg], acc[mg], 0, 0, 0);
                // (behind the MFMAs that read the register: they take their operands at issue, the data lands much later)
                gload_nt(wr[s], wsrc + (size_t)(cw * WIDE_KS + s) * 64);
            }
            // the other buffer was last read before the previous barrier
            if constexpr (PIECES == 8)
                asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xstage[0]), "+v"(xstage[1]), "+v"(xstage[2]), "+v"(xstage[3]), "+v"(xstage[4]), "+v"(xstage[5]), "+v"(xstage[6]), "+v"(xstage[7]) : "n"(X_WAIT));
            else if constexpr (PIECES == 6)
                asm volatile("s_waitcnt vmcnt(%6)" : "+v"(xstage[0]), "+v"(xstage[1]), "+v"(xstage[2]), "+v"(xstage[3]), "+v"(xstage[4]), "+v"(xstage[5]) : "n"(X_WAIT));
            else if constexpr (PIECES == 4)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xstage[0]), "+v"(xstage[1]), "+v"(xstage[2]), "+v"(xstage[3]) : "n"(X_WAIT));
            else
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xstage[0]), "+v"(xstage[1]) : "n"(X_WAIT));
#pragma unroll
            for (int i = 0; i < PIECES; ++i) xdst[i][(buf ^ 1) * BUF_STRIDE] = xstage[i];
            __syncthreads();
            buf ^= 1;
        };
        for (int c = c0; c < c1; c += 2) {
            chunk(c, wa, xa, xb);
            if (c + 1 < c1) chunk(c + 1, wb, xb, xa);   // (uniform)
        }
        // the loads still in flight write registers nobody reads any more: let them land before the registers are reused
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!rows_ok) return;
    if (p.partial == nullptr) {   // unsplit K: round and write the output here -- no partial tiles, no reduce launch
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            const int token = 32 * mg + (lane & 31);
            f32x4_t sum[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) sum[b] = f32x4_t{acc[mg][4 * b], acc[mg][4 * b + 1], acc[mg][4 * b + 2], acc[mg][4 * b + 3]};
            if (token < p.m_rows) wide_store(p.out, p.ld_out, n32, token, lane >> 5, sum, p.epilogue);
        }
        return;
    }
    if (p.token_major) {   // partial[split][token][32 n32 + 8 b + 4 h + c]: a consumer that works row by row reads it coalesced
        const int h = lane >> 5;
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            float *row = p.partial + ((size_t)split * (32 * MG) + 32 * mg + (lane & 31)) * ((size_t)p.n32 * 32) + 32 * n32 + 4 * h;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                *reinterpret_cast<f32x4_t *>(row + 8 * b) = f32x4_t{acc[mg][4 * b], acc[mg][4 * b + 1], acc[mg][4 * b + 2], acc[mg][4 * b + 3]};
        }
        return;
    }
    // D[row 8 b + 4 h + c][token j] = acc[4 b + c] of lane (h, j): stored as they are, 16 bytes per lane and b
    f32x4_t *out = reinterpret_cast<f32x4_t *>(p.partial) + (((size_t)split * p.n32 + n32) * MG) * 4 * 64 + lane;
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            out[(mg * 4 + b) * 64] = f32x4_t{acc[mg][4 * b], acc[mg][4 * b + 1], acc[mg][4 * b + 2], acc[mg][4 * b + 3]};
}

struct WideReduceParams {
    const float *partial;
    u16 *out;          // [m_rows, ld_out]
    int n32, splitk, mg_count, m_rows, ld_out, epilogue;
};

// One wave per (32 weight rows, token group): sums the splits in split order, rounds once, writes bf16.
// epilogue 1: the 32 rows are 16 gate rows then the 16 up rows of the same features (wide_gate_up_weight):
// out[token][16 tile + f] = silu(gate_f) * up_f.
__global__ __launch_bounds__(WIDE_THREADS) void wide_reduce_kernel(WideReduceParams p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = (int)blockIdx.x * 4 + wv;            // (n32, mg)
    if (item >= p.n32 * p.mg_count) return;
    const int n32 = item / p.mg_count, mg = item % p.mg_count;
    const int h = lane >> 5, token = 32 * mg + (lane & 31);
    f32x4_t sum[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) sum[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const size_t split_stride = (size_t)p.n32 * p.mg_count * 4 * 64;
    const f32x4_t *src = reinterpret_cast<const f32x4_t *>(p.partial) + ((size_t)n32 * p.mg_count + mg) * 4 * 64 + lane;
    for (int s = 0; s < p.splitk; ++s) {
#pragma unroll
        for (int b = 0; b < 4; ++b) sum[b] += src[(size_t)s * split_stride + b * 64];
    }
    if (token >= p.m_rows) return;
    wide_store(p.out, p.ld_out, n32, token, h, sum, p.epilogue);
}

// The residual add + RMSNorm that follows a split-K projection, reading its TOKEN-MAJOR partial tiles directly:
// delta = bf16(sum over the splits, in split order) -- the rounding of crag_enc_wide_reduce --, then crag_enc_rmsnorm's
// arithmetic (bf16 residual add, fp32 statistics, bf16(x * rstd) * w).  One workgroup per token row; replaces the reduce
// launch + the norm launch (about 5 us each at these sizes).
struct NormPartialsParams {
    const float *partial;   // [splitk][m_pad][hidden]
    const u16 *res_in, *w;
    u16 *out, *res_out;
    int splitk, m_pad, hidden, rows;
    float eps;
};

__global__ __launch_bounds__(WIDE_THREADS) void rmsnorm_partials_kernel(NormPartialsParams p) {
    __shared__ float sh[WIDE_THREADS / 64];
    const int r = blockIdx.x, tid = threadIdx.x;
    constexpr int MAXC = 4;                       // hidden <= 4 * 256 * 4 = 4096
    const int chunks = p.hidden >> 2;             // 4 columns (one f32x4 of a partial row) per step
    float v[MAXC][4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + i * WIDE_THREADS;
        if (c < chunks) {
            f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < p.splitk; ++s)
                sum += *reinterpret_cast<const f32x4_t *>(p.partial + ((size_t)s * p.m_pad + r) * p.hidden + 4 * c);
            const uint2 rr = *reinterpret_cast<const uint2 *>(p.res_in + (size_t)r * p.hidden + 4 * c);
            const u16 rin[4] = {(u16)(rr.x & 0xffffu), (u16)(rr.x >> 16), (u16)(rr.y & 0xffffu), (u16)(rr.y >> 16)};
            u16 s4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s4[e] = f2bf(bf2f(f2bf(sum[e])) + bf2f(rin[e]));   // delta rounded once, then the bf16 residual add
                v[i][e] = bf2f(s4[e]);
                ss += v[i][e] * v[i][e];
            }
            if (p.res_out)
                *reinterpret_cast<uint2 *>(p.res_out + (size_t)r * p.hidden + 4 * c) =
                    make_uint2((uint32_t)s4[0] | ((uint32_t)s4[1] << 16), (uint32_t)s4[2] | ((uint32_t)s4[3] << 16));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) sh[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < WIDE_THREADS / 64; ++i) tot += sh[i];
    const float rstd = rsqrtf(tot / (float)p.hidden + p.eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + i * WIDE_THREADS;
        if (c < chunks) {
            const uint2 ww = *reinterpret_cast<const uint2 *>(p.w + 4 * c);
            const u16 w4[4] = {(u16)(ww.x & 0xffffu), (u16)(ww.x >> 16), (u16)(ww.y & 0xffffu), (u16)(ww.y >> 16)};
            u16 o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(bf2f(w4[e]) * bf2f(f2bf(v[i][e] * rstd)));
            *reinterpret_cast<uint2 *>(p.out + (size_t)r * p.hidden + 4 * c) =
                make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        }
    }
}

// 128-row tiles (4 waves) unless they give the chip fewer than 100 workgroups: then 64-row tiles (2 waves, twice the
// workgroups at twice the activation bytes per byte of weight).  Measured (profiles/r04_wide_gemm.txt): down at 64 rows,
// splitk 4: 24.1 us with 80 workgroups of 128 rows, 18.9 with 160 of 64; gate|up unsplit: 29.0 us with 152 workgroups of
// 128 rows, 41.5 with 304 of 64.
void wide_launch(const WideParams &p, int m_pad, int n, hipStream_t st) {
    const int wg128 = (n / 128) * p.splitk;
    int tile = wg128 >= 100 ? 128 : 64;
    if (const char *v = getenv("CRAG_WIDE_TILE")) tile = atoi(v) == 64 ? 64 : 128;   // (developer switch)
    if (tile == 128) {
        const dim3 grid((unsigned)(n / 128), (unsigned)p.splitk);
        if (m_pad == 32) hipLaunchKernelGGL((wide_gemm_kernel<1, 4, 8>), grid, dim3(256), 0, st, p);
        else if (m_pad == 64) hipLaunchKernelGGL((wide_gemm_kernel<2, 4, 8>), grid, dim3(256), 0, st, p);
        else if (m_pad == 96) hipLaunchKernelGGL((wide_gemm_kernel<3, 4, 8>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wide_gemm_kernel<4, 4, 8>), grid, dim3(256), 0, st, p);
    } else {
        const dim3 grid((unsigned)(n / 64), (unsigned)p.splitk);
        if (m_pad == 32) hipLaunchKernelGGL((wide_gemm_kernel<1, 2, 4>), grid, dim3(128), 0, st, p);
        else if (m_pad == 64) hipLaunchKernelGGL((wide_gemm_kernel<2, 2, 4>), grid, dim3(128), 0, st, p);
        else if (m_pad == 96) hipLaunchKernelGGL((wide_gemm_kernel<3, 2, 4>), grid, dim3(128), 0, st, p);
        else hipLaunchKernelGGL((wide_gemm_kernel<4, 2, 4>), grid, dim3(128), 0, st, p);
    }
}

}  // namespace

extern "C" {

int64_t crag_enc_wide_partial_bytes(int m_pad, int n, int splitk) {
    if (m_pad <= 0 || n <= 0 || splitk <= 0) return -1;
    return (int64_t)splitk * n * m_pad * 4;
}

int crag_enc_wide_gemm(const uint16_t *x, const uint16_t *ww, float *partial, int m_pad, int n, int k, int splitk,
                       void *stream) {
    if (!x || !ww || !partial) return wfail("wide_gemm: NULL pointer");
    if (m_pad != 32 && m_pad != 64 && m_pad != 96 && m_pad != 128) return wfail("wide_gemm: m_pad must be 32, 64, 96 or 128 (got %d)", m_pad);
    if (n <= 0 || n % 128) return wfail("wide_gemm: n must be a multiple of 128 (got %d)", n);
    if (k <= 0 || k % WIDE_BK) return wfail("wide_gemm: k must be a multiple of %d (got %d)", WIDE_BK, k);
    if (splitk <= 0 || splitk > k / WIDE_BK) return wfail("wide_gemm: splitk must be in [1, k / %d] (got %d)", WIDE_BK, splitk);
    if (n % 64) return wfail("wide_gemm: n must be a multiple of 64");
    WideParams p;
    p.x = x;
    p.ww = ww;
    p.partial = partial;
    p.out = nullptr;
    p.m_rows = p.ld_out = p.epilogue = 0;
    p.K = k;
    p.n32 = n / 32;
    p.splitk = splitk;
    p.token_major = 0;
    wide_launch(p, m_pad, n, (hipStream_t)stream);
    return whip_ok("wide_gemm");
}

int crag_enc_wide_gemm_rows(const uint16_t *x, const uint16_t *ww, float *partial, int m_pad, int n, int k, int splitk,
                            void *stream) {
    if (!x || !ww || !partial) return wfail("wide_gemm_rows: NULL pointer");
    if (m_pad != 32 && m_pad != 64 && m_pad != 96 && m_pad != 128) return wfail("wide_gemm_rows: m_pad must be 32, 64, 96 or 128 (got %d)", m_pad);
    if (n <= 0 || n % 128) return wfail("wide_gemm_rows: n must be a multiple of 128 (got %d)", n);
    if (k <= 0 || k % WIDE_BK) return wfail("wide_gemm_rows: k must be a multiple of %d (got %d)", WIDE_BK, k);
    if (splitk <= 0 || splitk > k / WIDE_BK) return wfail("wide_gemm_rows: splitk must be in [1, k / %d] (got %d)", WIDE_BK, splitk);
    WideParams p;
    p.x = x;
    p.ww = ww;
    p.partial = partial;
    p.out = nullptr;
    p.m_rows = p.ld_out = p.epilogue = 0;
    p.K = k;
    p.n32 = n / 32;
    p.splitk = splitk;
    p.token_major = 1;
    wide_launch(p, m_pad, n, (hipStream_t)stream);
    return whip_ok("wide_gemm_rows");
}

int crag_enc_rmsnorm_partials(const float *partial_rows, int splitk, int m_pad, const uint16_t *residual_in,
                              const uint16_t *weight, uint16_t *out, uint16_t *residual_out, int rows, int hidden,
                              float eps, void *stream) {
    if (!partial_rows || !residual_in || !weight || !out) return wfail("rmsnorm_partials: NULL pointer");
    if (splitk <= 0 || rows <= 0 || rows > m_pad) return wfail("rmsnorm_partials: need splitk > 0 and 0 < rows <= m_pad");
    if (hidden <= 0 || hidden % 4 || hidden > 4096) return wfail("rmsnorm_partials: hidden must be a multiple of 4, at most 4096");
    NormPartialsParams p;
    p.partial = partial_rows;
    p.res_in = residual_in;
    p.w = weight;
    p.out = out;
    p.res_out = residual_out;
    p.splitk = splitk;
    p.m_pad = m_pad;
    p.hidden = hidden;
    p.rows = rows;
    p.eps = eps;
    hipLaunchKernelGGL(rmsnorm_partials_kernel, dim3((unsigned)rows), dim3(WIDE_THREADS), 0, (hipStream_t)stream, p);
    return whip_ok("rmsnorm_partials");
}

int crag_enc_wide_gemm_direct(const uint16_t *x, const uint16_t *ww, uint16_t *out, int m_rows, int m_pad, int n, int k,
                              int epilogue, void *stream) {
    if (!x || !ww || !out) return wfail("wide_gemm_direct: NULL pointer");
    if (m_pad != 32 && m_pad != 64 && m_pad != 96 && m_pad != 128) return wfail("wide_gemm_direct: m_pad must be 32, 64, 96 or 128 (got %d)", m_pad);
    if (m_rows <= 0 || m_rows > m_pad) return wfail("wide_gemm_direct: 0 < m_rows <= m_pad");
    if (n <= 0 || n % 128) return wfail("wide_gemm_direct: n must be a multiple of 128 (got %d)", n);
    if (k <= 0 || k % WIDE_BK) return wfail("wide_gemm_direct: k must be a multiple of %d (got %d)", WIDE_BK, k);
    if (epilogue != 0 && epilogue != 1) return wfail("wide_gemm_direct: epilogue must be 0 or 1");
    WideParams p;
    p.x = x;
    p.ww = ww;
    p.partial = nullptr;
    p.out = out;
    p.m_rows = m_rows;
    p.ld_out = epilogue ? n / 2 : n;
    p.epilogue = epilogue;
    p.K = k;
    p.n32 = n / 32;
    p.splitk = 1;
    p.token_major = 0;
    wide_launch(p, m_pad, n, (hipStream_t)stream);
    return whip_ok("wide_gemm_direct");
}

int crag_enc_wide_reduce(const float *partial, uint16_t *out, int m_rows, int m_pad, int n, int splitk, int epilogue,
                         void *stream) {
    if (!partial || !out) return wfail("wide_reduce: NULL pointer");
    if (m_pad != 32 && m_pad != 64 && m_pad != 96 && m_pad != 128) return wfail("wide_reduce: m_pad must be 32, 64, 96 or 128 (got %d)", m_pad);
    if (m_rows <= 0 || m_rows > m_pad) return wfail("wide_reduce: 0 < m_rows <= m_pad");
    if (n <= 0 || n % 128 || splitk <= 0) return wfail("wide_reduce: bad sizes n=%d splitk=%d", n, splitk);
    if (epilogue != 0 && epilogue != 1) return wfail("wide_reduce: epilogue must be 0 or 1");
    WideReduceParams p;
    p.partial = partial;
    p.out = out;
    p.n32 = n / 32;
    p.splitk = splitk;
    p.mg_count = m_pad / 32;
    p.m_rows = m_rows;
    p.ld_out = epilogue ? n / 2 : n;
    p.epilogue = epilogue;
    const int items = p.n32 * p.mg_count;
    hipLaunchKernelGGL(wide_reduce_kernel, dim3((unsigned)((items + 3) / 4)), dim3(WIDE_THREADS), 0, (hipStream_t)stream, p);
    return whip_ok("wide_reduce");
}

}  // extern "C"
