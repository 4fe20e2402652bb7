// crag_search.hip — exact cosine top-k over a 1024-d fp32 corpus for gfx950 (MI355X, CDNA4).
//
// Replaces the pgvector exact-scan lane of the reference
// (/root/reference/app/retrieve.py:326-389: `ORDER BY embedding <=> q LIMIT k`).
// Written for CDNA4 only: 64-wide waves, v_mfma_f32_32x32x2_f32 (exact fp32), buffer loads
// with hardware range checking, ds_swizzle cross-lane exchange.  See DESIGN.md for the layout
// and the roofline arithmetic.
//
// HBM layout ("tile32"): the corpus is stored in tiles of 32 rows; inside a tile a row is cut into 64 pieces of
// 16 dims (64 bytes), and piece p of row j lives at byte (p*32 + j)*64.  The float4 holding dims [4*kq, 4*kq+3] of
// row j is therefore at byte ((kq>>2)*32 + j)*64 + (kq&3)*16.  A wave-instruction `buffer_load_dwordx4` (lane l:
// row l&31, float4 2s + (l>>5) of the wave's K slice) reads one half of 32 consecutive pieces -- every 128-byte
// line it touches is completed by the next instruction -- and lands exactly in the B-operand lane map of
// v_mfma_f32_32x32x2_f32 (lane l: row l&31, k-half l>>5).  The pieces are what the exact rescoring of single rows
// (finalize_kernel) pays for: 64 lines per row.  [Float4-granular interleaving, piece = 16 bytes, made a scan
// instruction read 1 KiB of contiguous HBM but a single row touch 256 lines: 78 us of a 166 us top-100 search
// over 100 000 rows went into re-reading 8 400 rows.]
// Queries are normalised once per call into the same layout (the A operand).
//
// scan kernel: one 512-thread workgroup per CU; the 8 waves split K = 1024 into 8 slices of
// 128, each wave keeps its A slice (16 x float4) in registers and streams its 16 KiB slice of
// every tile of the workgroup's row range straight from HBM into registers (16 loads always
// in flight per wave, 128 KiB per CU).  Per tile the 8 partial 32x32 accumulators are summed
// through LDS; wave w then owns 4 of the 32 queries and keeps their running top-k in
// registers as half-wave (32-lane) sorted lists, updated with a bitonic sort/merge built on
// ds_swizzle.  Each workgroup writes its per-query top-k; merge_partials_kernel selects the
// final top-k (threshold pruning + rank-by-counting, radix select as the bounded fallback).
//
// These exact fp32 MFMA scans serve small corpora, rows with extreme norms and the overflow fallback.  From
// 128 rows per workgroup on, a search takes the prefilter path in the second half of this file: an fp16 MFMA
// scan over the fp16 mirror of the unit rows with a proven error bound, then exact fp32 rescoring of the few
// survivors from the fp32 rows (prefilter_kernel / finalize_kernel) -- same bits out, half the bytes in.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "crag_kernels.h"

namespace crag {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// a row piece = 2^PS float4 (template parameter PS of everything that touches the fp32 rows).  Two layouts:
//  * PS_SMALL = 2 (64-byte pieces) for an index WITHOUT the fp16 mirror, whose every search streams the fp32 rows: a scan
//    instruction reads half of each line it touches and the next one the other half.  [Measured against 3 and 4 on
//    one box: with 128-byte pieces a scan instruction reads a quarter of each line and the scan of the fp32 rows
//    slows from 73 to 83 us at 100 000 rows x 64 queries.]
//  * PS_BIG = 5 (512-byte pieces = a wave's whole K slice of a row) for an index WITH the mirror (the default): there
//    the fp32 rows are read by the exact rescoring of the prefilter path -- a survivor's row is then 8 x 512 contiguous
//    bytes in full 128-byte lines, where 64-byte pieces use half of every line they fetch: the 8 400 rows x 4 KiB of a
//    top-100 search over 64 queries were 10-16 us of a 28 us selection launch, HBM-bound at twice the useful bytes --
//    and by the fp32 scans of small corpora, irregular indices and the overflow fallback, which are latency-bound or
//    rare (and ~13 % slower per byte in this layout).
constexpr int PS_SMALL = 2, PS_BIG = 5;

// ------------------------------------------------------------------------------------------
// key helpers: a candidate is the 64-bit key (orderable(score) << 32) | ~row ; larger = better
// (higher score, then lower row position).  Key 0 is the "empty" sentinel.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t mk64(uint32_t hi, uint32_t lo) {
    return ((uint64_t)hi << 32) | lo;
}

// lane ^ X inside each 32-lane half (ds_swizzle bit mode: and=0x1f, or=0, xor=X)
template <int X>
__device__ __forceinline__ uint32_t swz_xor(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (X << 10) | 0x1f);
}

// compare-exchange with lane^X; `want_max` lanes keep the larger key, the others the smaller
template <int X>
__device__ __forceinline__ void cmpx(uint32_t &hi, uint32_t &lo, bool want_max) {
    const uint32_t phi = swz_xor<X>(hi), plo = swz_xor<X>(lo);
    const bool keep = (mk64(hi, lo) > mk64(phi, plo)) == want_max;
    hi = keep ? hi : phi;
    lo = keep ? lo : plo;
}

// sort the 32 keys of each half-wave, descending with lane index
__device__ __forceinline__ void sort32_desc(uint32_t &hi, uint32_t &lo, int lane) {
    const bool b1 = !(lane & 1), b2 = !(lane & 2), b4 = !(lane & 4), b8 = !(lane & 8),
               b16 = !(lane & 16);
    cmpx<1>(hi, lo, b1);
    cmpx<3>(hi, lo, b2);
    cmpx<1>(hi, lo, b1);
    cmpx<7>(hi, lo, b4);
    cmpx<2>(hi, lo, b2);
    cmpx<1>(hi, lo, b1);
    cmpx<15>(hi, lo, b8);
    cmpx<4>(hi, lo, b4);
    cmpx<2>(hi, lo, b2);
    cmpx<1>(hi, lo, b1);
    cmpx<31>(hi, lo, b16);
    cmpx<8>(hi, lo, b8);
    cmpx<4>(hi, lo, b4);
    cmpx<2>(hi, lo, b2);
    cmpx<1>(hi, lo, b1);
}

// sort a bitonic 32-sequence of each half-wave, descending
__device__ __forceinline__ void merge32_desc(uint32_t &hi, uint32_t &lo, int lane) {
    cmpx<16>(hi, lo, !(lane & 16));
    cmpx<8>(hi, lo, !(lane & 8));
    cmpx<4>(hi, lo, !(lane & 4));
    cmpx<2>(hi, lo, !(lane & 2));
    cmpx<1>(hi, lo, !(lane & 1));
}

// Running top-(32*S) of one query per half-wave: position p = s*32 + (lane & 31), descending.
template <int S>
struct HalfList {
    uint32_t hi[S], lo[S];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int s = 0; s < S; ++s) hi[s] = lo[s] = 0u;
    }
    // key currently at position k-1 of this lane's half (the k-th best so far; 0 if none)
    __device__ __forceinline__ uint64_t kth(int k, int lane) const {
        const int slot = (k - 1) >> 5, ln = (k - 1) & 31;
        uint32_t h = hi[0], l = lo[0];
#pragma unroll
        for (int s = 1; s < S; ++s)
            if (slot == s) {
                h = hi[s];
                l = lo[s];
            }
        const uint32_t h0 = __builtin_amdgcn_readlane(h, ln), h1 = __builtin_amdgcn_readlane(h, ln + 32);
        const uint32_t l0 = __builtin_amdgcn_readlane(l, ln), l1 = __builtin_amdgcn_readlane(l, ln + 32);
        return (lane & 32) ? mk64(h1, l1) : mk64(h0, l0);
    }
    // merge 32 new keys per half (one per lane, any order) into the list
    __device__ __forceinline__ void insert(uint32_t nhi, uint32_t nlo, int k, int lane) {
        if (!__any(mk64(nhi, nlo) > kth(k, lane))) return;  // wave-uniform: nothing beats the k-th
        sort32_desc(nhi, nlo, lane);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t rhi = swz_xor<31>(nhi), rlo = swz_xor<31>(nlo);  // carry, ascending
            const bool g = mk64(rhi, rlo) > mk64(hi[s], lo[s]);
            if (__any(g)) {  // wave-uniform; otherwise slot s and the carry are both unchanged
                const uint32_t mxh = g ? rhi : hi[s], mxl = g ? rlo : lo[s];
                const uint32_t mnh = g ? hi[s] : rhi, mnl = g ? lo[s] : rlo;
                hi[s] = mxh;
                lo[s] = mxl;
                merge32_desc(hi[s], lo[s], lane);
                if (s + 1 < S) {
                    nhi = mnh;
                    nlo = mnl;
                    merge32_desc(nhi, nlo, lane);
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// scan kernels
// ------------------------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Everything the two scan kernels share: row range, buffer window, A operand, owner map.
struct ScanCtx {
    int lane, w, g, qb, j, h;
    int64_t r_begin, r_end, t_begin;
    int n_tiles;
    bool reverse;
    uint32_t lane_off;
    int qloc[2];
    bool qok[2];
    const uint32_t *mrow[2];
};

// row range / lane geometry of one scan workgroup (everything but the query fields)
template <int PS>
__device__ __forceinline__ ScanCtx make_row_ctx(int64_t n_rows, int G, int reverse, int g, int qb) {
    ScanCtx c;
    c.lane = threadIdx.x & 63;
    c.w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    c.g = g;
    c.qb = qb;
    c.j = c.lane & 31;
    c.h = c.lane >> 5;
    // this workgroup's row range, balanced in units of 8 rows (one 128-B line per k-quad)
    const int64_t n8 = (n_rows + 7) >> 3;
    c.r_begin = ((n8 * c.g) / G) << 3;
    c.r_end = ((n8 * (c.g + 1)) / G) << 3;
    if (c.r_end > n_rows) c.r_end = n_rows;
    c.t_begin = c.r_begin >> 5;
    const int64_t t_end = (c.r_end > c.r_begin) ? ((c.r_end + 31) >> 5) : c.t_begin;
    c.n_tiles = (int)(t_end - c.t_begin);
    c.reverse = reverse != 0;
    c.lane_off = (uint32_t)(c.w * (KSLICE * 32 * 4) + c.j * (16 << PS) + c.h * 16);
    c.qloc[0] = c.qloc[1] = 0;
    c.qok[0] = c.qok[1] = false;
    c.mrow[0] = c.mrow[1] = nullptr;
    return c;
}

// (g, qb): the workgroup's position in the scan grid -- blockIdx.x / .y of a stand-alone scan kernel
template <int PS>
__device__ __forceinline__ ScanCtx make_ctx(const ScanParams &p, int g, int qb) {
    ScanCtx c = make_row_ctx<PS>(p.n_rows, p.G, p.reverse, g, qb);
    // after the LDS reduction this wave holds accumulator registers r = 2w, 2w+1 ->
    // query (r&3) + 8*(r>>2) + 4*h of the block, corpus row j of the tile
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * c.w + e;
        c.qloc[e] = (r & 3) + 8 * (r >> 2) + 4 * c.h;
        const int qglob = c.qb * 32 + c.qloc[e];
        c.qok[e] = qglob < p.nq;
        c.mrow[e] = p.mask ? p.mask + (size_t)(c.qok[e] ? qglob : 0) * (size_t)p.mask_stride_w : nullptr;
    }
    return c;
}

// byte offset of this lane's first load of tile `ti` (relative to the workgroup window), or an
// out-of-range marker when the lane's row is not ours / the tile does not exist: the buffer
// range check then returns zeros without touching memory
// `step` counts tiles in processing order; a reversed pass walks the range back to front so that
// the tail of the previous pass is still in the Infinity Cache (boustrophedon streaming)
__device__ __forceinline__ int tile_of(const ScanCtx &c, int step) { return c.reverse ? c.n_tiles - 1 - step : step; }

// byte offset, relative to tile_voff(), of a lane's s-th 16-byte load of a tile (s = 0..15 inside the wave's K
// slice): float4 kq = 2s + h of the slice sits in the 64-byte piece kq >> 2 of its row, at (kq & 3) * 16
template <int PS>
__device__ __forceinline__ constexpr uint32_t b_soff(int s) {
    return (uint32_t)(((2 * s) >> PS) * (512 << PS) + ((2 * s) & ((1 << PS) - 1)) * 16);
}

__device__ __forceinline__ uint32_t tile_voff(const ScanCtx &c, int step) {
    const int ti = tile_of(c, step);
    const int64_t row = (c.t_begin + ti) * 32 + c.j;
    const bool valid = (step < c.n_tiles) && (row >= c.r_begin) && (row < c.r_end);
    return valid ? (uint32_t)ti * (uint32_t)(TILE_FLOATS * 4) + c.lane_off : 0x80000000u;
}

// ---- the canonical 1/||q||: ONE piece of arithmetic for every kernel that needs a query's norm -------------------
// Thread t < 256 holds dims 4t .. 4t+3 of the query (zeros beyond dim / for threads >= 256); squares in fp64, a
// butterfly sum inside each of the first four waves, the four wave sums added in wave order.  prep_queries_kernel,
// the selection kernel and the fallback scan all call this, so the exact scores (x 1/||row|| x 1/||q||) are the same
// bits on every path.  All threads of the workgroup must call it (two barriers); sh: 4 doubles of LDS.
__device__ __forceinline__ f32x4 load_query_quad(const float *queries, int dim, int q, int nq, int t) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < nq && t < 256) {
        const float *src = queries + (size_t)q * dim;
        if ((dim & 3) == 0) {
            if (4 * t < dim) v = *reinterpret_cast<const f32x4 *>(src + 4 * t);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (4 * t + c < dim) v[c] = src[4 * t + c];
        }
    }
    return v;
}

__device__ __forceinline__ float canonical_qinv(const f32x4 v, bool real_query, double *sh) {
    double ss = ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0 && wv < 4) sh[wv] = ss;
    __syncthreads();
    const double tot = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    const bool ok = real_query && (tot > 0.0) && (tot < 1.0e300) && (tot == tot);
    float qinv = ok ? (float)(1.0 / sqrt(tot)) : 0.f;
    if (!(qinv < 3.0e38f)) qinv = 0.f;
    return qinv;
}

// A operand of the generic kernel: this wave's K slice of the (up to) 32 RAW queries of block qb; lane (i = lane&31,
// h = lane>>5) holds q[i][128w + 8s + 4h + 0..3] in a[s] -- the same k permutation the tile32 corpus layout gives
// the B operand.  qinv_s[32] (LDS) receives the canonical 1/||q|| of the block's queries (0: zero / non-finite /
// missing query, never eligible); sh: 4 doubles of LDS.  All threads of the workgroup call this.
__device__ __forceinline__ void load_queries(const ScanParams &p, const ScanCtx &c, f32x4 (&a)[16], float *qinv_s,
                                             double *sh) {
    const int qi = c.qb * 32 + c.j;
    const bool have = qi < p.nq;
    const float *qrow = p.queries + (size_t)(have ? qi : 0) * p.dim;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int col = c.w * KSLICE + 8 * s + 4 * c.h;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (have) {
            if ((p.dim & 3) == 0) {
                if (col < p.dim) v = *reinterpret_cast<const f32x4 *>(qrow + col);
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    if (col + cc < p.dim) v[cc] = qrow[col + cc];
            }
        }
        a[s] = v;
    }
    for (int i = 0; i < 32; ++i) {  // (32 x two barriers: this kernel is the rare path / the A-B baseline)
        const int q = c.qb * 32 + i;
        const float r = canonical_qinv(load_query_quad(p.queries, p.dim, q, p.nq, (int)threadIdx.x), q < p.nq, sh);
        if (threadIdx.x == 0) qinv_s[i] = r;
    }
    __syncthreads();
}

// score -> key for one owned (query, row) pair
// (the pipelined kernels' and the selection kernel's expression, so that all of them produce the same bits:
// scale = 1/||row|| * 1/||q||, or 0 when the pair is not eligible)
__device__ __forceinline__ void make_key(float dot, float scale, int64_t row, uint32_t &khi, uint32_t &klo) {
    float sc = dot * scale;
    sc = __builtin_amdgcn_fmed3f(sc, -1.f, 1.f);  // pgvector clamps the similarity to [-1, 1]
    const bool ok = (scale > 0.f) && (sc == sc);
    khi = ok ? f2ord(sc) : 0u;
    klo = ok ? ~(uint32_t)row : 0u;
}

template <int S>
__device__ __forceinline__ void write_lists(const ScanParams &p, const ScanCtx &c, const HalfList<S> (&list)[2]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (!c.qok[e]) continue;
        uint2 *dst = p.partial + (((size_t)c.qb * p.G + c.g) * 32 + c.qloc[e]) * (size_t)p.k;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int pos = s * 32 + c.j;
            if (pos < p.k) dst[pos] = make_uint2(list[e].hi[s], list[e].lo[s]);
        }
    }
}

#define CRAG_MFMA(A_, B_, ACC_) __builtin_amdgcn_mfma_f32_32x32x2f32((A_), __uint_as_float(B_), (ACC_), 0, 0, 0)

// ---- generic kernel (any S): MFMA phase, then reduction + selection, one barrier per tile ----
// The body is a device function so that the prefilter path's selection kernel can run it as its fallback (workgroups
// of the same launch, see finalize_fb_kernel): slab = 64 KiB of LDS, [buf][producer wave][reg pair][lane].
template <int S, int PS>
__device__ __forceinline__ void scan_body(const ScanParams &p, const int g, const int qb,
                                          float2 (*slab)[SCAN_WAVES][8][64]) {
    const ScanCtx c = make_ctx<PS>(p, g, qb);
    const int lane = c.lane, w = c.w, j = c.j;

    const float *wg_base = p.corpus + (size_t)c.t_begin * TILE_FLOATS;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wg_base), 0, (int)((uint32_t)c.n_tiles * (uint32_t)(TILE_FLOATS * 4)), 0x00020000);

    u32x4 b[16];
    {
        const uint32_t v0 = tile_voff(c, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v0 + b_soff<PS>(s), 0, 0);
    }
    f32x4 a[16];
    float qinv[2];
    {   // LDS scratch inside the (still unused) slab: 32 floats + 4 doubles
        float *qinv_s = reinterpret_cast<float *>(&slab[0][0][0][0]);
        load_queries(p, c, a, qinv_s, reinterpret_cast<double *>(qinv_s + 32));
        qinv[0] = qinv_s[c.qloc[0]];
        qinv[1] = qinv_s[c.qloc[1]];
        __syncthreads();  // the slab is about to be written
    }
    // drain with the compiler's own builtin so its vmcnt scoreboard is empty at the loop head:
    // otherwise the loop-head merge keeps a conservative wait on the A registers in every
    // iteration and drains the prefetch ring
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

    HalfList<S> list[2];
    list[0].clear();
    list[1].clear();

    int buf = 0;
    for (int ti = 0; ti < c.n_tiles; ++ti) {
        const uint32_t vnext = tile_voff(c, ti + 1);
        const int64_t tile = c.t_begin + tile_of(c, ti);
        const int64_t row = tile * 32 + j;
        const float inv_cur = p.inv_norm[row];  // row < cap_rows: the tile exists
        uint32_t mword[2] = {0xffffffffu, 0xffffffffu};
        if (p.mask) {
            mword[0] = c.mrow[0][tile];
            mword[1] = c.mrow[1][tile];
        }
        // 16 steps of {4 MFMA on b[s]; refill b[s] from the next tile}; the sched_barrier pins the
        // interleave so every load is issued one whole tile of MFMAs before its use
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc = CRAG_MFMA(a[s][0], b[s][0], acc);
            acc = CRAG_MFMA(a[s][1], b[s][1], acc);
            acc = CRAG_MFMA(a[s][2], b[s][2], acc);
            acc = CRAG_MFMA(a[s][3], b[s][3], acc);
            b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + b_soff<PS>(s), 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // split-K reduction through LDS (double-buffered: one barrier per tile)
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) slab[buf][w][pr][lane] = make_float2(acc[2 * pr], acc[2 * pr + 1]);
        __syncthreads();
        float d[2] = {0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < SCAN_WAVES; ++ww) {  // fixed order: bit-reproducible
            const float2 v = slab[buf][ww][w][lane];
            d[0] += v.x;
            d[1] += v.y;
        }
        buf ^= 1;
        const bool row_ok = (row >= c.r_begin) && (row < c.r_end) && (inv_cur > 0.f);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            uint32_t khi, klo;
            make_key(d[e], (row_ok && c.qok[e] && ((mword[e] >> j) & 1u)) ? inv_cur * qinv[e] : 0.f, row, khi, klo);
            list[e].insert(khi, klo, p.k, lane);
        }
    }
    write_lists<S>(p, c, list);
}

template <int S, int PS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(ScanParams p) {
    __shared__ float2 slab[2][SCAN_WAVES][8][64];
    scan_body<S, PS>(p, blockIdx.x, blockIdx.y, slab);
}

// Few candidates (the common case once the thresholds have tightened): insert them one by one into the
// sorted 32-entry half-wave list instead of running the 21-stage network.  Per round each half-wave
// takes its first remaining candidate, ranks it against the list with a ballot + popcount, and the
// tail of the list shifts down by one lane.  Data-dependent trip count (wave-uniform).
constexpr int SPARSE_MAX = 3;  // use this path when no half-wave has more candidates than this

__device__ __forceinline__ void sparse_insert(uint32_t &lh, uint32_t &ll, uint32_t kh, uint32_t kl, bool cand, int lane) {
    const bool hi_half = (lane & 32) != 0;
    const int pp = lane & 31;
    for (;;) {
        const uint64_t m = __ballot(cand);
        if (m == 0ull) break;
        const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
        const int s0 = m0 ? __builtin_ctz(m0) : 0, s1 = m1 ? 32 + __builtin_ctz(m1) : 32;
        const uint32_t ch0 = __builtin_amdgcn_readlane(kh, s0), cl0 = __builtin_amdgcn_readlane(kl, s0);
        const uint32_t ch1 = __builtin_amdgcn_readlane(kh, s1), cl1 = __builtin_amdgcn_readlane(kl, s1);
        const bool valid = hi_half ? (m1 != 0u) : (m0 != 0u);
        const uint32_t ch = hi_half ? ch1 : ch0, cl = hi_half ? cl1 : cl0;
        cand = cand && (lane != (hi_half ? s1 : s0));
        const uint64_t g = __ballot(mk64(lh, ll) > mk64(ch, cl));  // entries that stay ahead of the candidate
        const int pos = hi_half ? __popc((uint32_t)(g >> 32)) : __popc((uint32_t)g);
        const uint32_t uh = (uint32_t)__shfl_up((int)lh, 1, 32), ul = (uint32_t)__shfl_up((int)ll, 1, 32);
        if (valid && pp >= pos) {
            lh = (pp == pos) ? ch : uh;
            ll = (pp == pos) ? cl : ul;
        }
    }
}

// number of set bits of the fuller half of a wave mask
__device__ __forceinline__ int max_half_popc(uint64_t m) {
    const int a = __popc((uint32_t)m), b = __popc((uint32_t)(m >> 32));
    return a > b ? a : b;
}

// ---- pipelined kernel (k <= 32): the split-K reduction and the top-k selection of tile t-1 are
// cut into small ops (<= ~12 VALU each) and spread over the MFMA slots of tile t, so that the
// matrix pipe never waits for the LDS round trip or the sorting network.  The queries are used
// raw as the A operand; 1/||q|| is applied when a score becomes a key. ------------------------
struct PipeSel {  // both owned register lists (e = 0, 1) move through the network together
    uint32_t h[2], l[2];    // keys in flight (batch, later the merged bitonic sequence)
    uint32_t ph[2], pl[2];  // partner keys of the swizzle issued in the previous op
    uint32_t th[2], tl[2];  // current k-th key of each list, broadcast over its half-wave
    uint32_t pub[2];        // best score already published to the global bound (lane 0 of the half)
    bool active;            // wave-uniform: this batch has a key that beats a current k-th
};

template <int X>
__device__ __forceinline__ void net_issue(PipeSel &n) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        n.ph[e] = swz_xor<X>(n.h[e]);
        n.pl[e] = swz_xor<X>(n.l[e]);
    }
}
template <int BIT>
__device__ __forceinline__ void net_consume(PipeSel &n, int lane) {
    const bool want_max = !(lane & BIT);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const bool keep = (mk64(n.h[e], n.l[e]) > mk64(n.ph[e], n.pl[e])) == want_max;
        n.h[e] = keep ? n.h[e] : n.ph[e];
        n.l[e] = keep ? n.l[e] : n.pl[e];
    }
}

// the network as (xor distance, direction bit) per stage: 15 sort stages (0..14), then after the
// merge-split with the list 5 bitonic-merge stages (16..20)
template <int ST> struct SortStage;
#define CRAG_STAGE(ST_, X_, B_) \
    template <> struct SortStage<ST_> { static constexpr int X = X_, BIT = B_; };
CRAG_STAGE(0, 1, 1)
CRAG_STAGE(1, 3, 2)
CRAG_STAGE(2, 1, 1)
CRAG_STAGE(3, 7, 4)
CRAG_STAGE(4, 2, 2)
CRAG_STAGE(5, 1, 1)
CRAG_STAGE(6, 15, 8)
CRAG_STAGE(7, 4, 4)
CRAG_STAGE(8, 2, 2)
CRAG_STAGE(9, 1, 1)
CRAG_STAGE(10, 31, 16)
CRAG_STAGE(11, 8, 8)
CRAG_STAGE(12, 4, 4)
CRAG_STAGE(13, 2, 2)
CRAG_STAGE(14, 1, 1)
CRAG_STAGE(16, 16, 16)
CRAG_STAGE(17, 8, 8)
CRAG_STAGE(18, 4, 4)
CRAG_STAGE(19, 2, 2)
CRAG_STAGE(20, 1, 1)
#undef CRAG_STAGE

struct PipeTile {  // epilogue operands of the tile whose partial sums sit in the slab
    uint32_t nrow;      // ~row position (the low key word)
    float scale[2];     // inv_norm[row] * 1/||q_e||, or 0 when (row, query e) is not eligible
    uint32_t tauh[2];   // global lower bound on the k-th best score of query e (orderable bits)
};

struct PipeState {
    float2 rd[SCAN_WAVES];
    float d[2];
    HalfList<1> list[2];
    PipeSel n;
};

constexpr int PIPE_OPS = 28;

// after a list changed: refresh its k-th key and publish an improved head to the global bound
__device__ __forceinline__ void pipe_commit(const ScanParams &p, const ScanCtx &c, PipeState &st) {
    PipeSel &n = st.n;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint64_t t = st.list[e].kth(p.k, c.lane);
        n.th[e] = (uint32_t)(t >> 32);
        n.tl[e] = (uint32_t)t;
        const uint32_t head = st.list[e].hi[0];
        if ((c.lane & 31) == 0 && c.qok[e] && head > n.pub[e]) {
            n.pub[e] = head;  // only when the head improved: ~ln(rows) times per list
            (void)__hip_atomic_fetch_max(p.gbound + (size_t)(c.qb * 32 + c.qloc[e]) * GB_CELLS + (c.g % p.k), head,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// background op OP of the tile-(t-1) epilogue; placed after MFMA 2*OP of tile t
template <int OP>
__device__ __forceinline__ void pipe_bg(const ScanParams &p, const ScanCtx &c, float2 (*slab)[SCAN_WAVES][8][64],
                                        int rbuf, const PipeTile &pt, PipeState &st) {
    const int lane = c.lane;
    PipeSel &n = st.n;
    if constexpr (OP == 0) {
        __syncthreads();  // every wave's partial accumulators of the previous tile are in the slab
    } else if constexpr (OP == 1) {
#pragma unroll
        for (int ww = 0; ww < SCAN_WAVES; ++ww) st.rd[ww] = slab[rbuf][ww][c.w][lane];
    } else if constexpr (OP == 2) {  // fixed summation order: bit-reproducible
        st.d[0] = ((st.rd[0].x + st.rd[1].x) + st.rd[2].x) + st.rd[3].x;
        st.d[1] = ((st.rd[0].y + st.rd[1].y) + st.rd[2].y) + st.rd[3].y;
    } else if constexpr (OP == 3) {
        st.d[0] = (((st.d[0] + st.rd[4].x) + st.rd[5].x) + st.rd[6].x) + st.rd[7].x;
        st.d[1] = (((st.d[1] + st.rd[4].y) + st.rd[5].y) + st.rd[6].y) + st.rd[7].y;
    } else if constexpr (OP == 4 || OP == 5) {  // score -> key
        constexpr int e = OP - 4;
        float sc = st.d[e] * pt.scale[e];
        sc = __builtin_amdgcn_fmed3f(sc, -1.f, 1.f);  // pgvector clamps the similarity to [-1, 1]
        const bool ok = (pt.scale[e] > 0.f) && (sc == sc);
        const uint32_t u = __float_as_uint(sc);
        const uint32_t ord = u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
        n.h[e] = ok ? ord : 0u;
        n.l[e] = ok ? pt.nrow : 0u;
    } else if constexpr (OP == 6) {
        // a key matters only if it beats this workgroup's k-th AND is not below the global bound
        const bool b0 = (mk64(n.h[0], n.l[0]) > mk64(n.th[0], n.tl[0])) && (n.h[0] >= pt.tauh[0]);
        const bool b1 = (mk64(n.h[1], n.l[1]) > mk64(n.th[1], n.tl[1])) && (n.h[1] >= pt.tauh[1]);
        const uint64_t m0 = __ballot(b0), m1 = __ballot(b1);
        n.active = (m0 | m1) != 0ull;
        if (n.active) {
            const int c0 = max_half_popc(m0), c1 = max_half_popc(m1);
            if ((c0 > c1 ? c0 : c1) <= SPARSE_MAX) {  // few candidates: insert them directly, skip the network
                sparse_insert(st.list[0].hi[0], st.list[0].lo[0], n.h[0], n.l[0], b0, lane);
                sparse_insert(st.list[1].hi[0], st.list[1].lo[0], n.h[1], n.l[1], b1, lane);
                pipe_commit(p, c, st);
                n.active = false;
            } else {
                net_issue<SortStage<0>::X>(n);
            }
        }
    } else if constexpr (OP >= 7 && OP <= 20) {  // consume sort stage OP-7, issue sort stage OP-6
        if (n.active) {
            net_consume<SortStage<OP - 7>::BIT>(n, lane);
            net_issue<SortStage<OP - 6>::X>(n);
        }
    } else if constexpr (OP == 21) {  // batch sorted descending; reverse it for the merge-split
        if (n.active) {
            net_consume<SortStage<14>::BIT>(n, lane);
            net_issue<31>(n);
        }
    } else if constexpr (OP == 22) {  // merge-split: keep the elementwise max (bitonic), drop the min
        if (n.active) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool gt = mk64(n.ph[e], n.pl[e]) > mk64(st.list[e].hi[0], st.list[e].lo[0]);
                n.h[e] = gt ? n.ph[e] : st.list[e].hi[0];
                n.l[e] = gt ? n.pl[e] : st.list[e].lo[0];
            }
            net_issue<SortStage<16>::X>(n);
        }
    } else if constexpr (OP >= 23 && OP <= 26) {  // consume merge stage OP-7, issue the next
        if (n.active) {
            net_consume<SortStage<OP - 7>::BIT>(n, lane);
            net_issue<SortStage<OP - 6>::X>(n);
        }
    } else if constexpr (OP == 27) {
        if (n.active) {
            net_consume<SortStage<20>::BIT>(n, lane);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                st.list[e].hi[0] = n.h[e];
                st.list[e].lo[0] = n.l[e];
            }
            pipe_commit(p, c, st);
        }
    }
}

// max over the 32 lanes of each half-wave, result in every lane
__device__ __forceinline__ uint32_t half_max_u32(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1f);   // lane ^ 16
    return t > v ? t : v;
}

// min over the 32 lanes of each half-wave, result in every lane (DPP butterflies + one swizzle)
__device__ __forceinline__ uint32_t half_min_u32(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    v = t < v ? t : v;
    t = swz_xor<16>(v);
    return t < v ? t : v;
}

template <int PS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_pipe_kernel(ScanParams p) {
    __shared__ float2 slab[2][SCAN_WAVES][8][64];
    if (p.gate && *p.gate == 0u) return;  // fallback launch behind the prefilter path: nothing overflowed
    const ScanCtx c = make_ctx<PS>(p, blockIdx.x, blockIdx.y);
    const int lane = c.lane, w = c.w, j = c.j;

    const float *wg_base = p.corpus + (size_t)c.t_begin * TILE_FLOATS;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wg_base), 0, (int)((uint32_t)c.n_tiles * (uint32_t)(TILE_FLOATS * 4)), 0x00020000);

    const __amdgpu_buffer_rsrc_t gb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.gbound, 0, (int)((uint32_t)gridDim.y * 32u * GB_CELLS * 4u), 0x00020000);
    u32x4 b[16];
    {
        const uint32_t v0 = tile_voff(c, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v0 + b_soff<PS>(s), 0, 0);
    }
    // A operand: the raw queries in fragment order (prep_queries_kernel), lane (i = lane&31, h = lane>>5) holds
    // q[i][128w + 8s + 4h + 0..3]: one coalesced 1 KiB load per s
    f32x4 a[16];
    {
        const f32x4 *afrag = reinterpret_cast<const f32x4 *>(p.a32) + ((size_t)(c.qb * SCAN_WAVES + w) * 16) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = afrag[s * 64];
    }
    // drain with the compiler's own builtin so its vmcnt scoreboard is empty at the loop head:
    // otherwise the loop-head merge keeps a conservative wait on the A registers in every
    // iteration and drains the prefetch ring
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

    PipeState st;
    st.list[0].clear();
    st.list[1].clear();
    st.n.active = false;
    st.d[0] = st.d[1] = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) st.n.h[e] = st.n.l[e] = st.n.ph[e] = st.n.pl[e] = st.n.th[e] = st.n.tl[e] = st.n.pub[e] = 0u;
#pragma unroll
    for (int ww = 0; ww < SCAN_WAVES; ++ww) st.rd[ww] = make_float2(0.f, 0.f);
    PipeTile prev;  // "no previous tile": scale 0 => every key empty, batch inactive
    prev.nrow = 0u;
    prev.scale[0] = prev.scale[1] = 0.f;
    prev.tauh[0] = prev.tauh[1] = 0u;
    float qinv[2];  // 1/||q|| of the two owned queries of this half-wave (0: zero / non-finite query)
#pragma unroll
    for (int e = 0; e < 2; ++e) qinv[e] = c.qok[e] ? p.qinv[c.qb * 32 + c.qloc[e]] : 0.f;

    int wbuf = 0;  // slab buffer the tile now being multiplied will be written to
    for (int ti = 0; ti < c.n_tiles; ++ti) {
        const uint32_t vnext = tile_voff(c, ti + 1);
        const int64_t tile = c.t_begin + tile_of(c, ti);
        const int64_t row = tile * 32 + j;
        const float inv_row = p.inv_norm[row];  // row < cap_rows: the tile exists
        uint32_t mword[2] = {0xffffffffu, 0xffffffffu};
        if (p.mask) {
            mword[0] = c.mrow[0][tile];
            mword[1] = c.mrow[1][tile];
        }
        // global bound: k buckets per query, bucket b = best score any workgroup g with g % k == b
        // has published; k distinct rows score >= the smallest bucket, so nothing below it can be
        // in the final top-k.  Stale values only prune less.
        uint32_t gbv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // plain cached load on purpose (see scan_pipe2_kernel): stale values only prune less
            uint32_t off = (j < p.k && c.qok[e]) ? (uint32_t)((c.qb * 32 + c.qloc[e]) * GB_CELLS + j) * 4u : 0x80000000u;
            asm volatile("" : "+v"(off));  // opaque: the load must be re-issued every tile
            const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(gb_rsrc, off, 0, 0);
            gbv[e] = (off == 0x80000000u) ? 0xffffffffu : v;
        }
        PipeTile cur;
        cur.nrow = ~(uint32_t)row;
        cur.scale[0] = cur.scale[1] = 0.f;
        cur.tauh[0] = cur.tauh[1] = 0u;

        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        static_for<0, 64>([&](auto M) {
            constexpr int m = decltype(M)::value;
            constexpr int s = m >> 2, cc = m & 3;
            acc = CRAG_MFMA(a[s][cc], b[s][cc], acc);
            if constexpr (cc == 3) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + b_soff<PS>(s), 0, 0);
            if constexpr ((m & 1) == 0 && (m >> 1) < PIPE_OPS) {
                pipe_bg<(m >> 1)>(p, c, slab, wbuf ^ 1, prev, st);
            }
            if constexpr (m == 57) {  // eligibility of this tile's (row, query) pairs, used next tile
                const bool row_ok = (row >= c.r_begin) && (row < c.r_end) && (inv_row > 0.f);
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    cur.scale[e] = (row_ok && ((mword[e] >> j) & 1u)) ? inv_row * qinv[e] : 0.f;
            }
            if constexpr (m == 59 || m == 61) {
                constexpr int e = (m - 59) >> 1;
                cur.tauh[e] = half_min_u32(gbv[e]);
                // what our own bucket already holds: publishing is pointless unless we beat it
                const uint32_t mine = (uint32_t)__shfl((int)gbv[e], (lane & 32) | (c.g % p.k));
                st.n.pub[e] = mine > st.n.pub[e] ? mine : st.n.pub[e];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) slab[wbuf][w][pr][lane] = make_float2(acc[2 * pr], acc[2 * pr + 1]);
        wbuf ^= 1;
        prev = cur;
    }
    if (c.n_tiles > 0) {  // drain: epilogue of the last tile
        static_for<0, PIPE_OPS>([&](auto O) { pipe_bg<decltype(O)::value>(p, c, slab, wbuf ^ 1, prev, st); });
    }
    write_lists<1>(p, c, st.list);
}

// ---- pipelined kernel family with the selection state in LDS: scan_pipe2_kernel<KS, NQB> ----------
// NQB = 2: 64 queries per pass.  Intensity doubles (32 flop/B): the pass is bound by the fp32 matrix
// pipe, not HBM.  Each wave multiplies every loaded B fragment with TWO query blocks whose A fragments
// both stay in registers (128 VGPRs); to make room, everything the selection needs only occasionally
// (running lists, their k-th keys, per-tile scales and bounds) lives in LDS, and the split-K slab is
// single-buffered (two barriers per tile).
// NQB = 1: 32 queries per pass with the same LDS-resident state (double-buffered slab, one barrier per
// tile, 16-deep B ring): the HBM-bound pass for k > 32, which does not fit scan_pipe_kernel's registers.
// KS = 1, 2, 4: list slots of 32 keys per query (k <= 32 * KS).  For KS > 1 the rare dense batches are
// inserted inline (not spread over the MFMA slots), the common sparse ones one key at a time.
struct Pipe2State {
    float2 rd[4];           // partial sums being reduced
    float d[2];
    float sc[2];            // score scale of the two lists being processed (0 = not eligible)
    uint32_t tb[2];         // global bound of their queries
    uint32_t th[2], tl[2];  // their current k-th keys
    uint32_t nrow;          // ~row of the tile whose partial sums sit in the slab
    PipeSel n;
};

template <int NQB>
struct Pipe2Ctx {
    int qloc[2];
    bool qok[NQB][2];
    int qglob[NQB][2];
    int bucket;    // this workgroup's global-bound bucket
};

template <int KS, int NQB>
struct Pipe2Lds {
    float2 slab[3 - NQB][SCAN_WAVES][NQB][8][64];  // 64 KiB split-K partial sums [buf][producer][query block][pair][lane]
    uint2 list[SCAN_WAVES][NQB][2][KS][64];        // running top-k lists (keys), one per owned query
    uint2 tinfo[SCAN_WAVES][NQB][2][64];           // per (query, row): x = scale bits, y = global bound
    uint2 kth[SCAN_WAVES][NQB][2][2];              // k-th key of every list, per half-wave
    uint32_t pub[SCAN_WAVES][NQB][2][2];           // best score known to be in our global-bound bucket
    float qinv[SCAN_WAVES][NQB][2][2];             // 1/||q|| of the owned queries
};

// Sparse insertion into a KS-slot list held in registers (position s*32 + lane&31, descending): per round
// each half-wave takes its first remaining candidate, ranks it with ballots, and the tail shifts by one.
template <int KS>
__device__ __forceinline__ void sparse_insert_ks(HalfList<KS> &hl, uint32_t kh, uint32_t kl, bool cand, int lane) {
    const bool hi_half = (lane & 32) != 0;
    const int pp = lane & 31;
    for (;;) {
        const uint64_t m = __ballot(cand);
        if (m == 0ull) break;
        const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
        const int s0 = m0 ? __builtin_ctz(m0) : 0, s1 = m1 ? 32 + __builtin_ctz(m1) : 32;
        const uint32_t ch0 = __builtin_amdgcn_readlane(kh, s0), cl0 = __builtin_amdgcn_readlane(kl, s0);
        const uint32_t ch1 = __builtin_amdgcn_readlane(kh, s1), cl1 = __builtin_amdgcn_readlane(kl, s1);
        const bool valid = hi_half ? (m1 != 0u) : (m0 != 0u);
        const uint32_t ch = hi_half ? ch1 : ch0, cl = hi_half ? cl1 : cl0;
        cand = cand && (lane != (hi_half ? s1 : s0));
        int pos = 0;  // entries that stay ahead of the candidate
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint64_t g = __ballot(mk64(hl.hi[s], hl.lo[s]) > mk64(ch, cl));
            pos += hi_half ? __popc((uint32_t)(g >> 32)) : __popc((uint32_t)g);
        }
        static_for<0, KS>([&](auto I) {  // back to front: the carry into slot s is the OLD tail of slot s-1
            constexpr int s = KS - 1 - decltype(I)::value;
            uint32_t uh = (uint32_t)__shfl_up((int)hl.hi[s], 1, 32), ul = (uint32_t)__shfl_up((int)hl.lo[s], 1, 32);
            if constexpr (s > 0) {
                const uint32_t th = (uint32_t)__shfl((int)hl.hi[s - 1], 31, 32), tl = (uint32_t)__shfl((int)hl.lo[s - 1], 31, 32);
                uh = pp == 0 ? th : uh;
                ul = pp == 0 ? tl : ul;
            }
            const int gp = s * 32 + pp;
            if (valid && gp >= pos) {
                hl.hi[s] = (gp == pos) ? ch : uh;
                hl.lo[s] = (gp == pos) ? cl : ul;
            }
        });
    }
}

// store the two updated lists of query block QB, refresh their k-th keys, publish improved bound keys
template <int QB, int NQB, int KS>
__device__ __forceinline__ void pipe2_commit(const ScanParams &p, const ScanCtx &c, const Pipe2Ctx<NQB> &c2,
                                             Pipe2Lds<KS, NQB> &L, uint2 l0, uint2 l1) {
    const int lane = c.lane;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint2 v = e == 0 ? l0 : l1;
        L.list[c.w][QB][e][0][lane] = v;
        const int ln = (p.k - 1) & 31;  // k-th key of each half, broadcast through LDS
        if ((lane & 31) == ln) L.kth[c.w][QB][e][c.h] = v;
        uint32_t *pub = &L.pub[c.w][QB][e][c.h];
        if ((lane & 31) == 0 && c2.qok[QB][e] && v.x > *pub) {
            *pub = v.x;  // publish only improvements over what our bucket is known to hold
            (void)__hip_atomic_fetch_max(p.gbound + (size_t)c2.qglob[QB][e] * GB_CELLS + c2.bucket, v.x,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// KS > 1: one list (query block QB, owned query e) whose batch has candidates
template <int QB, int NQB, int KS>
__device__ __forceinline__ void pipe2_update_ks(const ScanParams &p, const ScanCtx &c, const Pipe2Ctx<NQB> &c2,
                                                Pipe2Lds<KS, NQB> &L, int e, uint32_t kh, uint32_t kl, bool cand,
                                                int max_cnt) {
    const int lane = c.lane;
    HalfList<KS> hl;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const uint2 v = L.list[c.w][QB][e][s][lane];
        hl.hi[s] = v.x;
        hl.lo[s] = v.y;
    }
    if (max_cnt <= 2 * SPARSE_MAX) {
        sparse_insert_ks<KS>(hl, kh, kl, cand, lane);
    } else {
        hl.insert(cand ? kh : 0u, cand ? kl : 0u, p.k, lane);
    }
    const int kslot = (p.k - 1) >> 5, ln = (p.k - 1) & 31;
    uint2 kv = make_uint2(hl.hi[0], hl.lo[0]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        L.list[c.w][QB][e][s][lane] = make_uint2(hl.hi[s], hl.lo[s]);
        if (s == kslot) kv = make_uint2(hl.hi[s], hl.lo[s]);
    }
    if ((lane & 31) == ln) L.kth[c.w][QB][e][c.h] = kv;
    // global bound for k > buckets: publish the key at rank pub_rank (< 32); buckets * (pub_rank + 1) >= k
    uint32_t *pub = &L.pub[c.w][QB][e][c.h];
    if ((lane & 31) == p.pub_rank && c2.qok[QB][e] && hl.hi[0] > *pub) {
        *pub = hl.hi[0];
        (void)__hip_atomic_fetch_max(p.gbound + (size_t)c2.qglob[QB][e] * GB_CELLS + c2.bucket, hl.hi[0],
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int OP, int QB, int NQB, int KS>
__device__ __forceinline__ void pipe2_bg(const ScanParams &p, const ScanCtx &c, const Pipe2Ctx<NQB> &c2,
                                         Pipe2Lds<KS, NQB> &L, int rbuf, Pipe2State &st) {
    const int lane = c.lane;
    PipeSel &n = st.n;
    if constexpr (OP == 0) {
        if constexpr (QB == 0) __syncthreads();  // every wave's partial accumulators of the previous tile are in the slab
    } else if constexpr (OP == 1) {
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) st.rd[ww] = L.slab[rbuf][ww][QB][c.w][lane];
    } else if constexpr (OP == 2) {  // fixed summation order: bit-reproducible
        st.d[0] = ((st.rd[0].x + st.rd[1].x) + st.rd[2].x) + st.rd[3].x;
        st.d[1] = ((st.rd[0].y + st.rd[1].y) + st.rd[2].y) + st.rd[3].y;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) st.rd[ww] = L.slab[rbuf][4 + ww][QB][c.w][lane];
    } else if constexpr (OP == 3) {
        st.d[0] = (((st.d[0] + st.rd[0].x) + st.rd[1].x) + st.rd[2].x) + st.rd[3].x;
        st.d[1] = (((st.d[1] + st.rd[0].y) + st.rd[1].y) + st.rd[2].y) + st.rd[3].y;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint2 t = L.tinfo[c.w][QB][e][lane];
            st.sc[e] = __uint_as_float(t.x);
            st.tb[e] = t.y;
            const uint2 kk = L.kth[c.w][QB][e][c.h];
            st.th[e] = kk.x;
            st.tl[e] = kk.y;
        }
    } else if constexpr (OP == 4 || OP == 5) {
        constexpr int e = OP - 4;
        float sc = st.d[e] * st.sc[e];
        sc = __builtin_amdgcn_fmed3f(sc, -1.f, 1.f);  // pgvector clamps the similarity to [-1, 1]
        const bool ok = (st.sc[e] > 0.f) && (sc == sc);
        const uint32_t u = __float_as_uint(sc);
        const uint32_t ord = u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
        n.h[e] = ok ? ord : 0u;
        n.l[e] = ok ? st.nrow : 0u;
    } else if constexpr (OP == 6) {
        const bool b0 = (mk64(n.h[0], n.l[0]) > mk64(st.th[0], st.tl[0])) && (n.h[0] >= st.tb[0]);
        const bool b1 = (mk64(n.h[1], n.l[1]) > mk64(st.th[1], st.tl[1])) && (n.h[1] >= st.tb[1]);
        const uint64_t m0 = __ballot(b0), m1 = __ballot(b1);
        if constexpr (KS == 1) {
            n.active = (m0 | m1) != 0ull;
            if (n.active) {
                const int c0 = max_half_popc(m0), c1 = max_half_popc(m1);
                if ((c0 > c1 ? c0 : c1) <= SPARSE_MAX) {  // few candidates: insert them directly, skip the network
                    uint2 l0 = L.list[c.w][QB][0][0][lane], l1 = L.list[c.w][QB][1][0][lane];
                    sparse_insert(l0.x, l0.y, n.h[0], n.l[0], b0, lane);
                    sparse_insert(l1.x, l1.y, n.h[1], n.l[1], b1, lane);
                    pipe2_commit<QB, NQB, KS>(p, c, c2, L, l0, l1);
                    n.active = false;
                } else {
                    net_issue<SortStage<0>::X>(n);
                }
            }
        } else {
            n.active = false;  // the network ops stay idle: batches are handled here, inline
            if (m0 != 0ull) pipe2_update_ks<QB, NQB, KS>(p, c, c2, L, 0, n.h[0], n.l[0], b0, max_half_popc(m0));
            if (m1 != 0ull) pipe2_update_ks<QB, NQB, KS>(p, c, c2, L, 1, n.h[1], n.l[1], b1, max_half_popc(m1));
        }
    } else if constexpr (KS > 1) {
        // nothing: see OP 6
    } else if constexpr (OP >= 7 && OP <= 20) {
        if (n.active) {
            net_consume<SortStage<OP - 7>::BIT>(n, lane);
            net_issue<SortStage<OP - 6>::X>(n);
        }
    } else if constexpr (OP == 21) {
        if (n.active) {
            net_consume<SortStage<14>::BIT>(n, lane);
            net_issue<31>(n);
        }
    } else if constexpr (OP == 22) {  // merge-split with the list: keep the elementwise max (bitonic)
        if (n.active) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint2 cur = L.list[c.w][QB][e][0][lane];
                const bool gt = mk64(n.ph[e], n.pl[e]) > mk64(cur.x, cur.y);
                n.h[e] = gt ? n.ph[e] : cur.x;
                n.l[e] = gt ? n.pl[e] : cur.y;
            }
            net_issue<SortStage<16>::X>(n);
        }
    } else if constexpr (OP >= 23 && OP <= 26) {
        if (n.active) {
            net_consume<SortStage<OP - 7>::BIT>(n, lane);
            net_issue<SortStage<OP - 6>::X>(n);
        }
    } else if constexpr (OP == 27) {
        if (n.active) {
            net_consume<SortStage<20>::BIT>(n, lane);
            pipe2_commit<QB, NQB, KS>(p, c, c2, L, make_uint2(n.h[0], n.l[0]), make_uint2(n.h[1], n.l[1]));
        }
    }
}

template <int KS, int NQB, int PS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_pipe2_kernel(ScanParams p) {
    __shared__ Pipe2Lds<KS, NQB> L;
    if (p.gate && *p.gate == 0u) return;  // fallback launch behind the prefilter path: nothing overflowed
    constexpr int RING = NQB == 2 ? 8 : 16;   // B prefetch ring depth (loads in flight per wave)
    constexpr int SLOTS = 64 * NQB;           // MFMAs per tile and wave
    const ScanCtx c = make_ctx<PS>(p, blockIdx.x, blockIdx.y);  // row range; its query fields are not used here
    const int lane = c.lane, w = c.w, j = c.j;
    Pipe2Ctx<NQB> c2;
    c2.bucket = c.g % p.nb;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * w + e;
        c2.qloc[e] = (r & 3) + 8 * (r >> 2) + 4 * c.h;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            c2.qglob[qb][e] = ((int)blockIdx.y * NQB + qb) * 32 + c2.qloc[e];
            c2.qok[qb][e] = c2.qglob[qb][e] < p.nq;
        }
    }
    const float *wg_base = p.corpus + (size_t)c.t_begin * TILE_FLOATS;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wg_base), 0, (int)((uint32_t)c.n_tiles * (uint32_t)(TILE_FLOATS * 4)), 0x00020000);

    const __amdgpu_buffer_rsrc_t gb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.gbound, 0, (int)((uint32_t)gridDim.y * (uint32_t)(NQB * 32) * GB_CELLS * 4u), 0x00020000);
    // B ring: slot s % RING serves steps s and s + RING
    u32x4 b[RING];
    uint32_t vcur = tile_voff(c, 0);
#pragma unroll
    for (int s = 0; s < RING; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vcur + b_soff<PS>(s), 0, 0);

    // A operand: the raw queries of the block(s) in fragment order (prep_queries_kernel), lane (i = lane&31, h)
    // holds q[i][128w + 8s + 4h + 0..3]; 1/||q|| (also prepared) is applied to the score later
    f32x4 a[NQB][16];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const f32x4 *afrag = reinterpret_cast<const f32x4 *>(p.a32) +
                             ((size_t)(((int)blockIdx.y * NQB + qb) * SCAN_WAVES + w) * 16) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[qb][s] = afrag[s * 64];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int s = 0; s < KS; ++s) L.list[w][qb][e][s][lane] = make_uint2(0u, 0u);
            L.tinfo[w][qb][e][lane] = make_uint2(0u, 0u);  // "no previous tile": nothing eligible
            if ((lane & 31) == 0) {
                L.kth[w][qb][e][c.h] = make_uint2(0u, 0u);
                L.pub[w][qb][e][c.h] = 0u;
                L.qinv[w][qb][e][c.h] = c2.qok[qb][e] ? p.qinv[c2.qglob[qb][e]] : 0.f;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): empty scoreboard at the loop head (see scan_kernel)

    Pipe2State st;
    st.n.active = false;
    st.d[0] = st.d[1] = 0.f;
    st.nrow = 0u;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        st.n.h[e] = st.n.l[e] = st.n.ph[e] = st.n.pl[e] = st.n.th[e] = st.n.tl[e] = st.n.pub[e] = 0u;
        st.sc[e] = 0.f;
        st.tb[e] = st.th[e] = st.tl[e] = 0u;
    }
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) st.rd[ww] = make_float2(0.f, 0.f);
    uint32_t cur_nrow = 0u;
    int wbuf = 0;  // NQB == 1: slab buffer the tile now being multiplied is written to

    // slot plan (NQB = 2 / 1): epilogue operands loaded at LOADS, written to LDS from TINFO on (odd slots),
    // second barrier (single-buffered slab only) after op 3 of the last query block
    constexpr int LOADS = NQB == 2 ? 81 : 41;
    constexpr int TINFO = SLOTS - 1 - 2 * (2 * NQB - 1);  // 121 / 61
    // NQB = 2: the slab writes are software-pipelined too.  Block 0's accumulator is complete after slot 119
    // and goes to the slab in the odd slots 121..127 (under block 1's last run); block 1's is written in slots
    // 1..7 of the NEXT tile (under block 0's first run, which starts from C = 0), so the matrix pipe never waits
    // for the accumulator -> LDS hand-over.  The background ops therefore start OPS0 slots into the tile.
    constexpr int OPS0 = NQB == 2 ? 10 : 0;
    constexpr int BAR2 = OPS0 + 2 * (PIPE_OPS + 4) + 1;
    const f32x16 zero16 = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) acc[qb] = zero16;

    for (int ti = 0; ti < c.n_tiles; ++ti) {
        const uint32_t vnext = tile_voff(c, ti + 1);
        const int64_t tile = c.t_begin + tile_of(c, ti);
        const int64_t row = tile * 32 + j;
        const float inv_row = p.inv_norm[row];
        uint32_t mword[NQB][2], gbv[NQB][2];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int e = 0; e < 2; ++e) mword[qb][e] = gbv[qb][e] = 0xffffffffu;
        st.nrow = cur_nrow;  // the tile whose sums are in the slab
        cur_nrow = ~(uint32_t)row;
        const int rbuf = NQB == 2 ? 0 : (wbuf ^ 1);

        static_for<0, SLOTS>([&](auto M) {
            constexpr int m = decltype(M)::value;
            if constexpr (NQB == 2) {
                // MFMA order: groups of 16 = two steps x {8 MFMAs on block 0, then 8 on block 1}.  Switching the
                // accumulator costs the matrix pipe ~15 cycles, so each accumulator is kept for runs of 8.
                constexpr int grp = m >> 4, r16 = m & 15, qb = r16 >> 3, s = 2 * grp + ((r16 >> 2) & 1), cc = r16 & 3,
                              slot = s & 7;
                if constexpr (m == 0 || m == 8) acc[qb] = CRAG_MFMA(a[qb][s][cc], b[slot][cc], zero16);  // first of the tile
                else acc[qb] = CRAG_MFMA(a[qb][s][cc], b[slot][cc], acc[qb]);
                if constexpr ((m & 1) == 1 && m < 8) {  // block 1 of the PREVIOUS tile -> slab (see OPS0)
                    constexpr int i = m >> 1;
                    L.slab[0][w][1][2 * i][lane] = make_float2(acc[1][4 * i], acc[1][4 * i + 1]);
                    L.slab[0][w][1][2 * i + 1][lane] = make_float2(acc[1][4 * i + 2], acc[1][4 * i + 3]);
                }
                if constexpr ((m & 1) == 1 && m >= 121) {  // block 0 of THIS tile -> slab, after the second barrier
                    constexpr int i = (m - 121) >> 1;
                    L.slab[0][w][0][2 * i][lane] = make_float2(acc[0][4 * i], acc[0][4 * i + 1]);
                    L.slab[0][w][0][2 * i + 1][lane] = make_float2(acc[0][4 * i + 2], acc[0][4 * i + 3]);
                }
                if constexpr (r16 == 15) {  // both ring slots of the group consumed: refill for steps +8
#pragma unroll
                    for (int ds = 0; ds < 2; ++ds) {
                        constexpr int base = 2 * grp;
                        const int st2 = base + ds;
                        if constexpr (base < 8) b[(base + ds) & 7] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vcur + b_soff<PS>(st2 + 8), 0, 0);
                        else b[(base + ds) & 7] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + b_soff<PS>(st2 - 8), 0, 0);
                    }
                }
            } else {
                constexpr int s = m >> 2, cc = m & 3;
                if constexpr (m == 0) acc[0] = CRAG_MFMA(a[0][s][cc], b[s][cc], zero16);
                else acc[0] = CRAG_MFMA(a[0][s][cc], b[s][cc], acc[0]);
                if constexpr (cc == 3) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + b_soff<PS>(s), 0, 0);
            }
            if constexpr ((m & 1) == 0 && m >= OPS0) {
                constexpr int o = (m - OPS0) >> 1;
                if constexpr (o < PIPE_OPS) pipe2_bg<o, 0, NQB, KS>(p, c, c2, L, rbuf, st);
                else if constexpr (NQB == 2 && o < 2 * PIPE_OPS) pipe2_bg<o - PIPE_OPS, NQB - 1, NQB, KS>(p, c, c2, L, rbuf, st);
            }
            if constexpr (NQB == 2 && m == BAR2) {
                // every wave has finished reading the slab (op 3 of query block 1): second barrier of the tile,
                // after which the accumulators of THIS tile may overwrite it
                __syncthreads();
            }
            if constexpr (m == LOADS) {  // epilogue operands of this tile, consumed from slot TINFO on
#pragma unroll
                for (int q2 = 0; q2 < NQB; ++q2)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (p.mask)
                            mword[q2][e] = p.mask[(size_t)(c2.qok[q2][e] ? c2.qglob[q2][e] : 0) * (size_t)p.mask_stride_w + tile];
                        {
                            // plain cached load on purpose: a device-coherent (sc1) load of these hot lines is
                            // slow and, loads returning in order, stalls the whole prefetch ring behind it.  A
                            // stale value only prunes less; the streaming traffic evicts the line every few tiles.
                            uint32_t off = (j < p.nb && c2.qok[q2][e]) ? (uint32_t)(c2.qglob[q2][e] * GB_CELLS + j) * 4u : 0x80000000u;
                            asm volatile("" : "+v"(off));  // opaque: the load must be re-issued every tile
                            const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(gb_rsrc, off, 0, 0);
                            gbv[q2][e] = (off == 0x80000000u) ? 0xffffffffu : v;
                        }
                    }
            }
            // epilogue operands of THIS tile -> LDS, one list per odd slot TINFO.. (the ops that still read the
            // previous tile's entries finished long before)
            if constexpr ((m & 1) == 1 && m >= TINFO) {
                constexpr int idx = (m - TINFO) >> 1, q2 = idx >> 1, e = idx & 1;
                const bool row_ok = (row >= c.r_begin) && (row < c.r_end) && (inv_row > 0.f);
                const float scale = (row_ok && ((mword[q2][e] >> j) & 1u)) ? inv_row * L.qinv[w][q2][e][c.h] : 0.f;
                const uint32_t tau = half_min_u32(gbv[q2][e]);
                L.tinfo[w][q2][e][lane] = make_uint2(__float_as_uint(scale), tau);
                if ((lane & 31) == c2.bucket) atomicMax(&L.pub[w][q2][e][c.h], gbv[q2][e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (NQB == 1) {
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) L.slab[wbuf][w][0][pr][lane] = make_float2(acc[0][2 * pr], acc[0][2 * pr + 1]);
        }
        wbuf ^= 1;
        vcur = vnext;
    }
    st.nrow = cur_nrow;
    if (c.n_tiles > 0) {  // drain: epilogue of the last tile
        if constexpr (NQB == 2) {  // its block-1 accumulator is still in registers
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) L.slab[0][w][1][pr][lane] = make_float2(acc[1][2 * pr], acc[1][2 * pr + 1]);
        }
        const int rbuf = NQB == 2 ? 0 : (wbuf ^ 1);
        static_for<0, PIPE_OPS>([&](auto O) { pipe2_bg<decltype(O)::value, 0, NQB, KS>(p, c, c2, L, rbuf, st); });
        if constexpr (NQB == 2)
            static_for<0, PIPE_OPS>([&](auto O) { pipe2_bg<decltype(O)::value, NQB - 1, NQB, KS>(p, c, c2, L, rbuf, st); });
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!c2.qok[qb][e]) continue;
            uint2 *dst = p.partial + (((size_t)((int)blockIdx.y * NQB + qb) * p.G + c.g) * 32 + c2.qloc[e]) * (size_t)p.k;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (s * 32 + j < p.k) dst[s * 32 + j] = L.list[w][qb][e][s][lane];
        }
}

// ------------------------------------------------------------------------------------------
// merge of the per-workgroup lists: one 256-thread workgroup per query
// ------------------------------------------------------------------------------------------
constexpr int MERGE_THREADS = 256;
constexpr int MERGE_CAP = 2048;       // candidates kept in LDS
constexpr int MERGE_RANK_MAX = 512;   // above this many candidates: radix-select first
constexpr int MERGE_HEADS = 1024;     // list heads kept in LDS for the head threshold
constexpr int CRAG_MAX_K_ = 128;      // = CRAG_MAX_K of include/crag_dense.h

__device__ __forceinline__ uint64_t key_of(uint2 v) { return mk64(v.x, v.y); }

// number of histogram entries strictly above bin `tid` (256 bins, 256 threads)
__device__ __forceinline__ int suffix_above(const int *hist, int *wave_tot, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    const int v = hist[tid];
    int incl = v;  // sum of bins tid .. end-of-wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int nb = __shfl_down(incl, o);
        if (lane + o < 64) incl += nb;
    }
    if (lane == 0) wave_tot[wv] = incl;
    __syncthreads();
    int above = incl - v;
    for (int ww = wv + 1; ww < MERGE_THREADS / 64; ++ww) above += wave_tot[ww];
    return above;
}

__device__ __forceinline__ void merge_partials_body(const MergeParams &p, const int q) {
    __shared__ uint64_t cand[MERGE_CAP];
    __shared__ uint64_t win[CRAG_MAX_K_];
    __shared__ __attribute__((aligned(16))) uint64_t heads[MERGE_HEADS + 8];
    __shared__ unsigned long long s_tau;
    __shared__ int s_cnt;
    __shared__ int hist[256];
    __shared__ int wave_tot[MERGE_THREADS / 64];
    __shared__ unsigned long long wave_max[MERGE_THREADS / 64];
    __shared__ int s_digit, s_need;

    const int tid = threadIdx.x;
    const int qb = q >> 5, ql = q & 31;
    const uint2 *base = p.partial + ((size_t)qb * p.G * 32 + ql) * (size_t)p.k;
    const size_t lstride = (size_t)32 * p.k;  // entries between consecutive workgroups' lists
    const int k = p.k, n_lists = p.G;
    const int total = n_lists * k;
    const bool use_heads = (n_lists >= k) && (n_lists <= MERGE_HEADS);

    if (tid == 0) {
        s_tau = 0ull;
        s_cnt = 0;
    }
    __syncthreads();

    // 1. two lower bounds on the final k-th key: the largest k-th key of any single list, and
    //    the k-th largest list head (k distinct entries are >= it).  Each thread owns the lists
    //    l = tid, tid + 256, ...; the first one's leading entries are fetched up front (independent
    //    loads) so that the walk in step 2 rarely needs a dependent round trip to memory.
    constexpr int LEAD = 4;
    uint64_t lead[LEAD];
#pragma unroll
    for (int i = 0; i < LEAD; ++i) lead[i] = 0ull;
    {
        unsigned long long m = 0ull;
        for (int l = tid; l < n_lists; l += MERGE_THREADS) {
            const uint2 *lp = base + (size_t)l * lstride;
            const unsigned long long kk = key_of(lp[k - 1]);
            m = kk > m ? kk : m;
            if (l == tid) {
#pragma unroll
                for (int i = 0; i < LEAD; ++i)
                    if (i < k) lead[i] = key_of(lp[i]);
                if (use_heads) heads[l] = lead[0];
            } else if (use_heads) {
                heads[l] = key_of(lp[0]);
            }
        }
        if (use_heads && tid < 8 && n_lists + tid < MERGE_HEADS + 8) heads[n_lists + tid] = 0ull;
        // block max without 256-way contention on one LDS word
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(m, o);
            m = other > m ? other : m;
        }
        if ((tid & 63) == 0) wave_max[tid >> 6] = m;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = wave_max[0];
        for (int i = 1; i < MERGE_THREADS / 64; ++i) m = wave_max[i] > m ? wave_max[i] : m;
        s_tau = m;
    }
    __syncthreads();
    if (use_heads) {
        const int n_pad = (n_lists + 7) & ~7;  // heads[n_lists .. n_pad) were zeroed above
        const ulonglong2 *h2 = reinterpret_cast<const ulonglong2 *>(heads);
        for (int l = tid; l < n_lists; l += MERGE_THREADS) {
            const uint64_t mine = heads[l];
            if (mine == 0ull) continue;
            int rank = 0;
            for (int i = 0; i < n_pad / 2; i += 4) {  // 8 keys per iteration, 4 independent LDS reads
                const ulonglong2 a0 = h2[i], a1 = h2[i + 1], a2 = h2[i + 2], a3 = h2[i + 3];
                rank += (a0.x > mine) + (a0.y > mine) + (a1.x > mine) + (a1.y > mine) + (a2.x > mine) +
                        (a2.y > mine) + (a3.x > mine) + (a3.y > mine);
            }
            if (rank == k - 1) atomicMax(&s_tau, (unsigned long long)mine);
        }
        __syncthreads();
    }
    const uint64_t tau = s_tau;

    // 2. compact the survivors into LDS: lists are sorted, so walk each one only while >= tau
    for (int l = tid; l < n_lists; l += MERGE_THREADS) {
        const uint2 *lp = base + (size_t)l * lstride;
        for (int pos = 0; pos < k; ++pos) {
            uint64_t kk;
            if (l == tid && pos < LEAD) {
                kk = lead[0];
#pragma unroll
                for (int i = 1; i < LEAD; ++i)
                    if (pos == i) kk = lead[i];
            } else {
                kk = key_of(lp[pos]);
            }
            if (kk == 0ull || kk < tau) break;
            const int idx = atomicAdd(&s_cnt, 1);
            if (idx < MERGE_CAP) cand[idx] = kk;
        }
    }
    __syncthreads();
    int C = s_cnt;
    const bool in_lds = C <= MERGE_CAP;
    const uint64_t *src = cand;

    // 3. (rare) too many survivors for rank-by-counting: radix-select the k-th key, keep winners
    if (C > MERGE_RANK_MAX) {
        uint64_t prefix = 0ull, pmask = 0ull;
        int need = k;
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            if (in_lds) {
                for (int e = tid; e < C; e += MERGE_THREADS) {
                    const uint64_t kk = cand[e];
                    if ((kk & pmask) == prefix) atomicAdd(&hist[(int)((kk >> shift) & 255ull)], 1);
                }
            } else {
                for (int e = tid; e < total; e += MERGE_THREADS) {
                    const int l = e / k, pos = e - l * k;
                    const uint64_t kk = key_of(base[(size_t)l * lstride + pos]);
                    if (kk >= tau && kk != 0ull && (kk & pmask) == prefix)
                        atomicAdd(&hist[(int)((kk >> shift) & 255ull)], 1);
                }
            }
            __syncthreads();
            {  // digit d with  sum_{d'>d} hist < need <= sum_{d'>=d} hist
                const int above = suffix_above(hist, wave_tot, tid);
                const int here = hist[tid];
                if (above < need && above + here >= need) {
                    s_digit = tid;
                    s_need = need - above;
                }
            }
            __syncthreads();
            prefix |= (uint64_t)s_digit << shift;
            pmask |= 255ull << shift;
            need = s_need;
            __syncthreads();
        }
        // prefix is now the k-th largest key: gather the winners (exactly k of them, keys unique)
        const uint64_t kth = prefix;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (in_lds) {
            for (int e = tid; e < C; e += MERGE_THREADS) {
                const uint64_t kk = cand[e];
                if (kk >= kth) {
                    const int idx = atomicAdd(&s_cnt, 1);
                    if (idx < CRAG_MAX_K_) win[idx] = kk;
                }
            }
        } else {
            for (int e = tid; e < total; e += MERGE_THREADS) {
                const int l = e / k, pos = e - l * k;
                const uint64_t kk = key_of(base[(size_t)l * lstride + pos]);
                if (kk != 0ull && kk >= kth) {
                    const int idx = atomicAdd(&s_cnt, 1);
                    if (idx < CRAG_MAX_K_) win[idx] = kk;
                }
            }
        }
        __syncthreads();
        C = s_cnt < CRAG_MAX_K_ ? s_cnt : CRAG_MAX_K_;
        src = win;
    }

    // 4. rank by counting among the C candidates; rank < k goes to output slot `rank`
    const int count = C < k ? C : k;
    for (int e = tid; e < C; e += MERGE_THREADS) {
        const uint64_t mine = src[e];
        int rank = 0;
        for (int i = 0; i < C; ++i) rank += (src[i] > mine) ? 1 : 0;
        if (rank < k) {
            const uint32_t row = ~(uint32_t)(mine & 0xffffffffull);
            p.out_scores[(size_t)q * k + rank] = ord2f((uint32_t)(mine >> 32));
            p.out_ids[(size_t)q * k + rank] = p.ids ? p.ids[row] : (int64_t)row + p.id_base;
        }
    }
    for (int r = count + tid; r < k; r += MERGE_THREADS) {
        p.out_scores[(size_t)q * k + r] = __uint_as_float(0x7fc00000u);
        p.out_ids[(size_t)q * k + r] = -1;
    }
    if (tid == 0) p.out_counts[q] = count;
    // leave the global-bound cells of this query zeroed for the next scan that uses this workspace
    if (p.gbound && tid < GB_CELLS) p.gbound[(size_t)q * GB_CELLS + tid] = 0u;
}

__global__ __launch_bounds__(MERGE_THREADS) void merge_partials_kernel(MergeParams p) { merge_partials_body(p, blockIdx.x); }

// ------------------------------------------------------------------------------------------
// cross-shard merge (multi-GPU exchange step): [n_lists, nq, k] (ids, scores) -> [nq, k]
// order: score desc, id asc.  One 256-thread workgroup per query, rank by counting.
// ------------------------------------------------------------------------------------------
constexpr int XMERGE_CAP = 4096;

__global__ __launch_bounds__(MERGE_THREADS) void merge_results_kernel(XMergeParams p) {
    __shared__ uint32_t s_sc[XMERGE_CAP];
    __shared__ int64_t s_id[XMERGE_CAP];
    __shared__ int s_cnt;
    const int q = blockIdx.x, tid = threadIdx.x, k = p.k;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int e = tid; e < p.n_lists * k; e += MERGE_THREADS) {
        const int l = e / k, pos = e - l * k;
        // the count, the score and the id are requested together (slots beyond a list's count hold its -1 / NaN padding:
        // read and dropped) -- one memory round trip instead of count -> values
        const size_t src = (size_t)q * k + pos;
        const int cnt = p.counts[(size_t)l * p.stride_counts + q];
        const float sc = p.scores[(size_t)l * p.stride_scores + src];
        const int64_t id = p.ids[(size_t)l * p.stride_ids + src];
        if (pos < cnt) {
            const int idx = atomicAdd(&s_cnt, 1);
            s_sc[idx] = f2ord(sc);
            s_id[idx] = id;
        }
    }
    __syncthreads();
    const int C = s_cnt;
    const int count = C < k ? C : k;
    for (int e = tid; e < C; e += MERGE_THREADS) {
        const uint32_t ms = s_sc[e];
        const int64_t mi = s_id[e];
        int rank = 0;
        for (int i = 0; i < C; ++i) {
            const uint32_t os = s_sc[i];
            rank += (os > ms || (os == ms && s_id[i] < mi)) ? 1 : 0;
        }
        if (rank < k) {
            p.out_scores[(size_t)q * k + rank] = ord2f(ms);
            p.out_ids[(size_t)q * k + rank] = mi;
        }
    }
    for (int r = count + tid; r < k; r += MERGE_THREADS) {
        p.out_scores[(size_t)q * k + r] = __uint_as_float(0x7fc00000u);
        p.out_ids[(size_t)q * k + r] = -1;
    }
    if (tid == 0) p.out_counts[q] = count;
}

// ------------------------------------------------------------------------------------------
// layout kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wv] = v;
    __syncthreads();
    const double t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// rows [n, dim] row-major -> tile32 layout at row positions [pos, pos+n); also 1/||row||.
// One 256-thread block per row: thread kq moves dims [4kq, 4kq+3].
// With a mirror (nullable): the unit row rounded to fp16, in the order the prefilter scan's MFMA wants its B
// operand -- [tile][K slice w][k-step t8][lane (h, j)][8 halves], the halves being dims 128w + 16 t8 + 8 (e >> 2) +
// 4h + (e & 3) -- computed exactly as the scan would on the fly (fp32 multiply by 1/||row||, v_cvt_pk_f16_f32), so a
// scan of the mirror sees bit for bit the operand a scan of the fp32 rows builds in registers.
template <int PS>
__global__ __launch_bounds__(256) void store_rows_kernel(const float *rows, int dim, int64_t pos,
                                                         float *corpus, float *inv_norm, uint32_t *irregular,
                                                         _Float16 *mirror) {
    __shared__ double sh[4];
    __shared__ float sh_inv;
    const int64_t i = blockIdx.x;
    const int kq = threadIdx.x;
    const float *src = rows + (size_t)i * dim;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (4 * kq + 3 < dim && (dim & 3) == 0) {
        v = *reinterpret_cast<const f32x4 *>(src + 4 * kq);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * kq + c < dim) v[c] = src[4 * kq + c];
    }
    const int64_t row = pos + i;
    float *dst = corpus + (size_t)(row >> 5) * TILE_FLOATS + ((size_t)(kq >> PS) * 32 + (row & 31)) * (4 << PS) + (kq & ((1 << PS) - 1)) * 4;
    *reinterpret_cast<f32x4 *>(dst) = v;
    double ss = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    ss = block_sum_256(ss, sh);
    if (kq == 0) {
        // zero or non-finite norm (NaN/Inf anywhere in the row) => never eligible
        const bool ok = (ss > 0.0) && (ss < 1.0e300) && (ss == ss);
        float inv = ok ? (float)(1.0 / sqrt(ss)) : 0.f;
        if (!(inv > 0.f) || !(inv < 3.0e38f)) inv = 0.f;
        inv_norm[row] = inv;
        sh_inv = inv;
        // a norm far outside fp32's comfortable range: the index is kept off the fp16 prefilter path
        if (inv > 0.f && (inv < 1.0e-30f || inv > 1.0e30f)) *irregular = 1u;
    }
    if (mirror) {
        __syncthreads();
        const float inv = sh_inv;
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x4 u = v * inv;
        const f16x2_t lo = __builtin_convertvector((f32x2_t{u[0], u[1]}), f16x2_t);
        const f16x2_t hi = __builtin_convertvector((f32x2_t{u[2], u[3]}), f16x2_t);
        f16x4_t o = f16x4_t{lo[0], lo[1], hi[0], hi[1]};
        if (!(inv > 0.f)) {  // never eligible: NaN scores, whatever the query
            const _Float16 qnan = __builtin_bit_cast(_Float16, (unsigned short)0x7e00);
            o = f16x4_t{qnan, qnan, qnan, qnan};
        }
        const int w = kq >> 5, d = (4 * kq) & 127, t8 = d >> 4, g = (d >> 3) & 1, h = (d >> 2) & 1;
        const size_t idx = ((((size_t)(row >> 5) * SCAN_WAVES + w) * 8 + t8) * 64 + h * 32 + (size_t)(row & 31)) * 8 + 4 * g;
        *reinterpret_cast<f16x4_t *>(mirror + idx) = o;
    }
}

template <int PS>
__global__ __launch_bounds__(256) void load_rows_kernel(const float *corpus, int dim, int64_t pos,
                                                        float *rows) {
    const int64_t i = blockIdx.x;
    const int kq = threadIdx.x;
    const int64_t row = pos + i;
    const float *src = corpus + (size_t)(row >> 5) * TILE_FLOATS + ((size_t)(kq >> PS) * 32 + (row & 31)) * (4 << PS) + (kq & ((1 << PS) - 1)) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(src);
    float *dst = rows + (size_t)i * dim;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (4 * kq + c < dim) dst[4 * kq + c] = v[c];
}

__global__ __launch_bounds__(256) void count_eligible_kernel(const float *inv_norm, int64_t n,
                                                             const uint32_t *mask,
                                                             unsigned long long *out) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        bool ok = inv_norm[i] > 0.f;
        if (ok && mask) ok = (mask[i >> 5] >> (i & 31)) & 1u;
        c += ok ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

__global__ void fill_ids_kernel(int64_t *ids, int64_t pos, int64_t n, int64_t first) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[pos + i] = first + i;
}

// ==========================================================================================
// Prefilter path: fp16 MFMA scan (HBM-bound for any batch) + exact fp32 rescoring of the survivors
// ==========================================================================================
// The fp32 MFMA runs at 1/16 of the fp16 rate, which makes a 64-query pass matrix-pipe bound (4.2).  Here
// the corpus is streamed ONCE as unit rows rounded to fp16 -- read ready-made from the fp16 mirror that
// store_rows_kernel keeps beside the fp32 rows (2 KiB per row), or, for an index without the mirror, converted in
// registers from the fp32 rows (x 1/||row||, v_cvt_pk_f16_f32: the same bits) -- and multiplied with the fp16
// unit queries by v_mfma_f32_32x32x16_f16.  The result is the cosine up to a PROVEN bound delta (rounding both
// operands to 11 significant bits + fp32 accumulation, see PF_DELTA); a row whose approximate score is within
// 2*delta of a lower bound on the k-th best approximate score may be in the exact top-k and becomes a
// candidate, every other row cannot.  finalize_kernel rescoring the (few) candidates from the fp32 rows with the
// fixed-order fp32 chain of the scan kernels above gives bit-identical scores and order.  Bytes: N*D*2 (mirror)
// or N*D*4 (fp32 rows) of corpus stream + 4 KiB per rescored row (reported by bench.py).
//
// Lower bound on the k-th best approximate score of a query without any sorted list: rows are split into
// 32*sets classes (row position mod 32 = the lane that owns the row, x tile index mod sets); gbound[q][class]
// is the best approximate score seen in the class by ANY workgroup (atomic max).  The (k_s)-th largest of the 32
// class maxima of set s is attained by k_s distinct rows; with sum k_s = k, the minimum over the sets is a
// valid bound.  Stale reads only prune less.
constexpr float PF_DELTA = 1.25e-3f;
// |approx - cos| <= sum_i |q_i c_i| (2u + u^2) [u = 2^-11, fp16 round-to-nearest of both operands, each in
// [-1, 1] after normalisation]  <= 9.8e-4 by Cauchy-Schwarz, + GRADUAL UNDERFLOW of components below 2^-14 (absolute
// rounding error <= 2^-25 per element: 2 * 32 * 2^-25 = 1.9e-6), + fp32 accumulation of 1024 exact products and
// 7 partial sums, <= 1031 * 2^-23 = 1.23e-4 even if every add truncates, + the fp32 normalisations 4 * 2^-24; sum
// 1.105e-3 (the difference between the fp32 chain and the real-number cosine, < 2e-6, included).
// The underflow term is a HARDWARE assumption: v_cvt_pk_f16_f32 (mirror, query fragments) and
// v_mfma_f32_32x32x16_f16 must keep fp16 subnormals -- flushed to zero they would cost up to
// ||small part|| * ||q|| <= 32 * 2^-14 = 1.95e-3 > delta.  tests/test_subnormal_bound.py builds exactly that case
// (unit vectors whose mass sits in [0.85, 0.98] * 2^-14, on the row side and on the query side, planted around the
// k-th score so that a flushing implementation provably loses them -- the CPU half of the test shows it on a model
// of the rule) and the kernels pass it on MI355X / ROCm 7.2: subnormals are kept (ISA 7.4: MFMA never flushes C/D,
// A/B follow MODE.denorm, which hipcc leaves at "keep" for f16).  tests/test_prefilter_gpu.py measures the
// actual worst case on ordinary inputs.
constexpr int PF_FLUSH_ABOVE = 768;                // staged candidates that trigger a sift at the next tile boundary
constexpr int PF_STAGE = PF_FLUSH_ABOVE + 2048;    // per-workgroup staging entries in LDS: a tile adds at most 32 x 64
constexpr int PF_PER = (PF_STAGE + SCAN_THREADS - 1) / SCAN_THREADS;   // staged entries per thread in a sift
constexpr int PF_TAU_CELL = 128;                   // cell of a query's bound record that holds the DERIVED bound
constexpr int PF_STASH_MIRROR = 4;                 // tiles scored before the first bound can have arrived (mirror scan)
constexpr int PF_STASH_ROWS = 2;                   // ... scan of the fp32 rows: its tiles take twice as long (the exchange
                                                   // needs TIME, not tiles: lags 1 / 2), and its registers are full
// (the lags between a publish, the delegates' derivation and every wave's read -- 2 and 4 tiles -- are PfParams fields)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- K0: per query 1/||q||, the query in fp32 A-fragment order (raw) and fp16 A-fragment order (unit) --------
// One 256-thread workgroup per padded query; thread t owns dims 4t .. 4t+3.  (The per-query prefilter state --
// class maxima, candidate count -- is left zeroed by the selection kernel of the previous search; the overflow
// word holds the sequence number of the search that overflowed and needs no reset.)
__global__ __launch_bounds__(256) void prep_queries_kernel(PrepParams p) {
    __shared__ double sh[4];
    const int q = blockIdx.x, t = threadIdx.x;
    const f32x4 v = load_query_quad(p.queries, p.dim, q, p.nq, t);
    const float qinv = canonical_qinv(v, q < p.nq, sh);
    if (t == 0) p.qinv[q] = qinv;
    const int qb = q >> 5, i = q & 31, w = t >> 5, d = (4 * t) & 127;
    {   // fp32 fragments: float4 (s, h) of wave slice w, s = d/8, h = (d/4)&1
        const int sidx = d >> 3, h = (d >> 2) & 1;
        reinterpret_cast<f32x4 *>(p.a32)[((size_t)(qb * SCAN_WAVES + w) * 16 + sidx) * 64 + h * 32 + i] = v;
    }
    if (p.a16) {  // fp16 fragments of the unit query: k-step t8 = d/16, elements 4g..4g+3 (g = (d/8)&1) of lane (i, h)
        const int t8 = d >> 4, g = (d >> 3) & 1, h = (d >> 2) & 1;
        f32x2 lo = {v[0] * qinv, v[1] * qinv}, hi = {v[2] * qinv, v[3] * qinv};
        const f16x2 l2 = __builtin_convertvector(lo, f16x2), h2 = __builtin_convertvector(hi, f16x2);
        _Float16 *dst = p.a16 + ((((size_t)(qb * SCAN_WAVES + w) * 8 + t8) * 64 + h * 32 + i) * 8 + 4 * g);
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f16x4 *>(dst) = f16x4{l2[0], l2[1], h2[0], h2[1]};
    }
}

// lane ^ X inside each 32-lane half on the VALU (DPP) where a pattern exists (ds_swizzle costs an LDS round trip per
// stage: 20 us of a 690 us scan went into the 15 stages of the sort below).  scripts/probes/dpp_xor_check.hip
// checks every pattern against ds_swizzle on the device.
template <int X>
__device__ __forceinline__ uint32_t dpp_xor(uint32_t v, int lane) {
    if constexpr (X == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (X == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (X == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
    else if constexpr (X == 7) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (X == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true); // row_mirror
    else if constexpr (X == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
    else if constexpr (X == 4) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x12C, 0xF, 0xF, true);  // row_ror:12: lane i <- i+4
        const uint32_t dn = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, true);  // row_ror:4:  lane i <- i-4
        return (lane & 4) ? dn : up;
    } else return swz_xor<X>(v);
}

// sort the 32 values of each half-wave, descending with the lane index
__device__ __forceinline__ uint32_t sort32_desc_u32(uint32_t v, int lane) {
#define CRAG_CX(X_, BIT_)                                              \
    {                                                                  \
        const uint32_t o_ = dpp_xor<X_>(v, lane);                      \
        const bool mx_ = !(lane & BIT_);                               \
        v = mx_ ? (o_ > v ? o_ : v) : (o_ < v ? o_ : v);               \
    }
    CRAG_CX(1, 1) CRAG_CX(3, 2) CRAG_CX(1, 1) CRAG_CX(7, 4) CRAG_CX(2, 2) CRAG_CX(1, 1) CRAG_CX(15, 8) CRAG_CX(4, 4)
    CRAG_CX(2, 2) CRAG_CX(1, 1) CRAG_CX(31, 16) CRAG_CX(8, 8) CRAG_CX(4, 4) CRAG_CX(2, 2) CRAG_CX(1, 1)
#undef CRAG_CX
    return v;
}

// Whole-wave reductions on the VALU: DPP inside the 16-lane rows, v_permlane16_swap / v_permlane32_swap across them
// (a ds_bpermute shuffle is an LDS-pipeline round trip, ~100 cycles each with one wave on the SIMD: the 100 dependent
// shuffles of a k-th-value search were 4 us).  Every lane ends with the result.
template <class F>
__device__ __forceinline__ uint32_t wave_reduce_u32(uint32_t v, int lane, F f) {
    v = f(v, dpp_xor<1>(v, lane));
    v = f(v, dpp_xor<2>(v, lane));
    v = f(v, dpp_xor<4>(v, lane));
    v = f(v, dpp_xor<8>(v, lane));
    const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // rows 0<->1, 2<->3
    v = f(r16[0], r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // the two halves
    return f(r32[0], r32[1]);
}

template <int NQB>
struct PfLds {
    typedef float slab_t __attribute__((ext_vector_type(2 * NQB)));
    slab_t slab[2][SCAN_WAVES][SCAN_WAVES][64];  // [buf][owner wave][producer wave][lane]: split-K partial sums
    uint2 stage[PF_STAGE];                       // staged candidates: x = orderable score, y = (row - window) << 6 | query
    float qthr[32 * NQB];                        // the owners' current candidate thresholds (they only rise)
    uint32_t qcount[32 * NQB], qbase[32 * NQB], qfill[32 * NQB];
    uint32_t n_stage;
    uint32_t need[NQB];                          // bit q: a wave found no bound for query q of the pass (emergency derivation)
    uint32_t want_flush[2][SCAN_WAVES];          // [slab buffer][wave]: the wave saw the staging buffer fill up
};

// What a wave owns after the split-K reduction: RPO = 2*NQB queries per lane (row j = lane & 31), consecutive
// indices: register R = w*RPO + e of the concatenated accumulators is query (R>>4)*32 + (r&3) + 8*(r>>2) + 4h
// with r = R & 15, and w*RPO is a multiple of RPO, so (r & 3) = e (+2 for odd w when RPO = 2).
template <int NQB>
struct PfOwner {
    int ql0;            // first owned query inside the pass; the others are ql0 + e
    int qg0;            // ... and as a global query index
    uint32_t okmask;    // bit e: owned query e is a real query with a finite non-zero norm
};

// candidates of one tile -> LDS staging.  sc: approximate cosines (NaN = not eligible), pass: candidate predicate
template <int NQB>
__device__ __forceinline__ void pf_stage(PfLds<NQB> &L, const PfOwner<NQB> &o, const float (&sc)[2 * NQB],
                                         const bool (&pass)[2 * NQB], uint32_t row_in_window, uint32_t *flags, uint32_t seq) {
#pragma unroll
    for (int e = 0; e < 2 * NQB; ++e) {
        if (pass[e]) {
            const uint32_t slot = atomicAdd(&L.n_stage, 1u);
            if (slot < (uint32_t)PF_STAGE) {
                L.stage[slot] = make_uint2(f2ord(sc[e]), (row_in_window << 6) | (uint32_t)(o.ql0 + e));
            } else {
                *flags = seq;  // cannot happen (<= PF_FLUSH_ABOVE staged + <= 2048 per tile); never drop one silently
            }
        }
    }
}

// The sift (all threads of the workgroup; contains barriers).  A staged candidate passed the threshold its query had
// when it was scored; thresholds only rise, and most of what an early, weak bound let through fails the current one.
// So every staged candidate is re-tested against its query's CURRENT threshold (L.qthr) and the survivors are packed
// in place; only if the buffer is still more than half full afterwards (or at the workgroup's end: LAST) do they move
// to the per-query global lists -- one returning global atomic per workgroup, query and flush.  [Before: whatever was
// staged went to the global lists as it was, 2 900 candidates per query of a top-100 search over 1M rows where the
// final bound passes ~400; the selection kernel paid for them.]
template <int NQB, bool LAST>
__device__ __forceinline__ void pf_sift(const PfParams &p, PfLds<NQB> &L, int64_t window_row0) {
    __syncthreads();   // every append and threshold update of the tile is visible
    const uint32_t n = L.n_stage < (uint32_t)PF_STAGE ? L.n_stage : (uint32_t)PF_STAGE;
    const int tid = threadIdx.x;
    if (LAST && n == 0u) return;  // (uniform)
    uint2 ent[PF_PER];
    bool keep[PF_PER];
#pragma unroll
    for (int i = 0; i < PF_PER; ++i) {
        const uint32_t idx = (uint32_t)tid + (uint32_t)i * SCAN_THREADS;
        keep[i] = false;
        if (idx < n) {
            ent[i] = L.stage[idx];
            keep[i] = ord2f(ent[i].x) >= L.qthr[ent[i].y & 63u];
        }
    }
    if (tid < 32 * NQB) {
        L.qcount[tid] = 0u;
        L.qfill[tid] = 0u;
    }
    __syncthreads();   // every entry is in registers
    if (tid == 0) L.n_stage = 0u;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF_PER; ++i)
        if (keep[i]) {
            atomicAdd(&L.qcount[ent[i].y & 63u], 1u);
            if (!LAST) L.stage[atomicAdd(&L.n_stage, 1u)] = ent[i];
        }
    __syncthreads();
    if (!LAST && L.n_stage <= (uint32_t)(PF_FLUSH_ABOVE / 2)) return;  // (uniform) the survivors stay staged
    if (tid < 32 * NQB) {
        const uint32_t cnt = L.qcount[tid];
        const int qg = (int)blockIdx.y * (32 * NQB) + tid;
        L.qbase[tid] = cnt ? atomicAdd(&p.count[qg], cnt) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF_PER; ++i)
        if (keep[i]) {
            const uint32_t ql = ent[i].y & 63u;
            const uint32_t pos = L.qbase[ql] + atomicAdd(&L.qfill[ql], 1u);
            const int qg = (int)blockIdx.y * (32 * NQB) + (int)ql;
            if (pos < (uint32_t)p.cap) {
                p.cand[(size_t)qg * p.cap + pos] = make_uint2(ent[i].x, (uint32_t)(window_row0 + (int64_t)(ent[i].y >> 6)));
            } else {
                p.flags[0] = p.seq;  // this query's list is full: the fallback blocks of the selection launch take over
            }
        }
    if (LAST) return;
    __syncthreads();
    if (tid == 0) L.n_stage = 0u;
    __syncthreads();
}

// ---- K1: the fp16 scan.  Grid (G, passes); one pass = 32*NQB queries against this workgroup's row range ------
// A lane keeps the maxima of ITS rows (class = lane x tile parity: the rows of a class within a workgroup all belong
// to one lane) in registers, and the workgroups exchange them on a geometric schedule only: publish after tiles 0,
// 3, 15, 63, ... and once more four tiles before the end (atomic max of the class maxima that rose since the last
// checkpoint and beat the bound).  A bound derived from the first t tiles of every workgroup lets about 12 n/t rows
// per query pass (k = 10, n tiles per workgroup) and holds for the next 3t tiles, so every interval adds the same
// few dozen candidates per query.  All other tiles touch no shared word.
//
// Who derives the bound (round 4).  Until round 3 every wave derived the bounds of the queries it owns: it read
// their 32 * SETS class maxima and sorted them, RPO * SETS half-wave sorts per wave and exchange -- 16 per wave at
// k = 100, three exchanges in a 100 000-row scan: ~13 us of a 52 us scan.  Now workgroup g is the DELEGATE of query
// g mod (32 * NQB) of its pass: two tiles after a publish its wave 0 reads that one query's class maxima, sorts
// them (SETS / 2 sorts: the two half-waves take two sets at a time) and publishes the result -- min over the sets
// of the k_s-th largest class maximum -- into the query's bound cell (atomic max; with 256 workgroups and 64
// queries four delegates per query write the same value).  Every wave reads the bound cells of the queries it owns
// two tiles later: one load per query instead of a sort per query and set.  The price is a first bound that arrives
// after tile 4 instead of tile 2: the first PF_STASH = 4 tiles' scores wait in registers and are judged last.
// A wave that still finds no published bound at its end derives one itself, the old way (a delegate that is far
// behind must not send a whole search into the fallback).
//
// What happens to candidates of early, weak bounds: see pf_sift.  The thresholds of the owned queries live in LDS
// (L.qthr) besides the owners' registers so that the sift can re-test any staged candidate.
// [The first version exchanged on every tile: bound loads in front of the MFMA phase, a sort, an atomic max for
// every row that beat its class.  Switching its parts off on one box (100 000 rows x 64 queries / 1M x 64): nothing
// exchanged and no candidates 64.6 / 641 us; + bound loads of lines nobody writes 66.2 / 651; + the atomic maxima
// 91.3 / 658; + sorts and candidate staging 104.5 / 690.  A device-scope atomic is carried out at the memory side
// and acknowledged microseconds later; it sits in the same in-order queue as the corpus loads of its wave, a
// stalled wave stalls its workgroup at the next barrier, and while bounds are young some wave of every workgroup
// lifts a class maximum on nearly every tile.  Plain stores instead of atomics ran at 78 / 626 us but lose
// maxima (20x the candidates); eight copies of the cells (one per XCD) weaken the bound 10x for no gain.]

// publish points of a workgroup with n tiles: 0, then read_lag + 1 (the first publish that can be filtered by a
// bound: an unfiltered one is an atomic from every lane for every owned query and set -- 1.5 M of them on 8192 cells
// at k = 100), 4 P + 3 from there on, then n - 4 (the bound every workgroup reads at its end and sifts its staged
// candidates with); a geometric point within 25 % of the last one is skipped.  Returns a value >= 2^30 when there is
// none left.
__device__ __forceinline__ int pf_next_pub(int after, int n_tiles, int read_lag) {
    const int last = n_tiles >= 12 ? n_tiles - 4 : -1;
    const int np = after < 0 ? 0 : (after == 0 ? read_lag + 1 : after * 4 + 3);
    if (after < last && np + (np >> 2) >= last) return last;
    return (np < n_tiles) ? np : (1 << 30);
}

template <int NQB, int SETS, bool MIRROR, bool NT>
__global__ __launch_bounds__(SCAN_THREADS) void prefilter_kernel(PfParams p) {
    constexpr int RPO = 2 * NQB;
    // Cache policy of the corpus loads: NT = streaming ("slc" / nt: the lines are not kept) for a mirror far larger
    // than the 256 MB Infinity Cache -- 1M rows x 64: 4-9 % faster (348 -> 335 us on one box, 340 -> 310 on another).
    // Up to 500 000 rows it makes no difference or hurts: a 100 000-row mirror (205 MB) stays cached between
    // searches (the passes alternate direction) and is 3 us FASTER with plain loads.  Never for the scan of the
    // fp32 rows, whose load instructions each use half of the lines they touch and count on finding the other
    // half cached (83 vs 71 us at 100 000 rows, 705 vs 660 us at 1M).
    constexpr int AUX = NT ? 2 : 0;
    __shared__ PfLds<NQB> L;
    // Prologue order matters (a scan of 100 000 rows is only ~13 tiles per workgroup): the first tile's corpus loads
    // and the query fragments leave before anything waits on memory; the owned queries' norms (one vector load)
    // ride behind them.  [Before: four dependent qinv loads, each waited for, and two 64-bit divisions stood in
    // front of the first corpus load.]
    ScanCtx c;
    c.lane = threadIdx.x & 63;
    c.w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    c.g = blockIdx.x;
    c.qb = blockIdx.y;
    c.j = c.lane & 31;
    c.h = c.lane >> 5;
    c.reverse = p.reverse != 0;
    constexpr int PS = PS_SMALL;   // (the fp32 rows are scanned here only for an index WITHOUT the mirror)
    c.lane_off = (uint32_t)(c.w * (KSLICE * 32 * 4) + c.j * (16 << PS) + c.h * 16);
    {   // whole tiles per workgroup: tiles [nt*g/G, nt*(g+1)/G) in 32-bit arithmetic (nt < 2^27, G < 2^16)
        const uint32_t nt = (uint32_t)((p.n_rows + 31) >> 5), G = (uint32_t)p.G, g = (uint32_t)c.g;
        const uint32_t qt = nt / G, rt = nt - qt * G;
        const uint32_t t0 = qt * g + (rt * g) / G, t1 = qt * (g + 1) + (rt * (g + 1)) / G;
        c.t_begin = t0;
        c.n_tiles = (int)(t1 - t0);
        c.r_begin = (int64_t)t0 * 32;
        c.r_end = (int64_t)t1 * 32 < p.n_rows ? (int64_t)t1 * 32 : p.n_rows;
    }
    const int lane = c.lane, w = c.w, j = c.j, h = c.h;

    // B operand source.  fp32 rows (tile32 layout): two 16-byte loads per k-step, normalised and rounded to fp16
    // in registers.  fp16 mirror (store_rows_kernel): one 16-byte load per k-step, already the MFMA operand --
    // half the bytes per row; 64 KiB per tile, this wave's K slice = 8 KiB, a k-step = 1 KiB across the wave.
    constexpr int NB = MIRROR ? 8 : 16;                       // loads per lane and tile
    constexpr uint32_t TILE_BYTES = MIRROR ? TILE_FLOATS * 2 : TILE_FLOATS * 4;
    const char *wg_base = MIRROR ? reinterpret_cast<const char *>(p.corpus16) + (size_t)c.t_begin * TILE_BYTES
                                 : reinterpret_cast<const char *>(p.corpus) + (size_t)c.t_begin * TILE_BYTES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(wg_base), 0, (int)((uint32_t)c.n_tiles * TILE_BYTES), 0x00020000);
    const uint32_t lane_off = MIRROR ? (uint32_t)(w * (8 * 1024) + lane * 16) : c.lane_off;
    // byte offset of this lane's first load of the tile of step `step`, or the out-of-range marker (reads zeros)
    auto voff = [&](int step) -> uint32_t {
        const int t = tile_of(c, step);
        const int64_t r = (c.t_begin + t) * 32 + j;
        const bool valid = (step < c.n_tiles) && (r >= c.r_begin) && (r < c.r_end);
        return valid ? (uint32_t)t * TILE_BYTES + lane_off : 0x80000000u;
    };
    u32x4 b[NB];
    {
        const uint32_t v0 = voff(0);
#pragma unroll
        for (int s = 0; s < NB; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v0 + (MIRROR ? (uint32_t)s * 1024u : b_soff<PS>(s)), 0, AUX);
    }
    // A operand: fp16 unit queries in fragment order: lane (i, h) holds for k-step t8 the dims
    // 128w + 16 t8 + 8 (e >> 2) + 4h + (e & 3), e = 0..7 -- the order two consecutive B loads deliver
    f16x8 a[NQB][8];
    PfOwner<NQB> o;
    {
        const int R = w * RPO, qb = R >> 4, r = R & 15;
        o.ql0 = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        o.qg0 = (int)blockIdx.y * (32 * NQB) + o.ql0;
        o.okmask = 0u;
    }
    {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const f16x8 *af = reinterpret_cast<const f16x8 *>(p.a16) +
                              ((size_t)(((int)blockIdx.y * NQB + qb) * SCAN_WAVES + w) * 8) * 64 + lane;
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) a[qb][t8] = af[t8 * 64];
        }
        // the RPO owned queries are consecutive and qg0 is a multiple of RPO: one aligned vector load (qinv is
        // nq_pad long; padded queries hold 0)
        typedef float qv_t __attribute__((ext_vector_type(RPO)));
        const qv_t qi = *reinterpret_cast<const qv_t *>(p.qinv + o.qg0);
#pragma unroll
        for (int e = 0; e < RPO; ++e)
            if (o.qg0 + e < p.nq && qi[e] > 0.f) o.okmask |= 1u << e;
    }
    if (threadIdx.x < 32 * NQB) L.qthr[threadIdx.x] = -__builtin_inff();
    if (threadIdx.x < NQB) L.need[threadIdx.x] = 0u;
    if (threadIdx.x == 0) L.n_stage = 0u;
    // the first PF_STASH tiles' scores (NaN = not eligible): scored before any bound can have arrived, judged at the end
    constexpr int PF_STASH = MIRROR ? PF_STASH_MIRROR : PF_STASH_ROWS;
    float stash[PF_STASH][RPO];   // (their row positions are recomputed at the end: registers are the scarce thing here)
#pragma unroll
    for (int t = 0; t < PF_STASH; ++t)
#pragma unroll
        for (int e = 0; e < RPO; ++e) stash[t][e] = __uint_as_float(0x7fc00000u);
    float thr[RPO];    // candidate thresholds of the owned queries (bound - 2 delta)
    uint32_t tau[RPO];            // the bound itself, orderable (0 = none yet)
    uint32_t lmax[RPO][SETS];     // class maxima over this lane's own rows
    uint32_t dirty = 0u;          // bit e*SETS+s: lmax[e][s] has risen since the last publish
#pragma unroll
    for (int e = 0; e < RPO; ++e) {
        thr[e] = -__builtin_inff();
        tau[e] = 0u;
#pragma unroll
        for (int s = 0; s < SETS; ++s) lmax[e][s] = 0u;
    }
    float inv_cur = 1.f;  // (mirror: rows that may never match are NaN in the mirror itself)
    if (!MIRROR && c.n_tiles > 0) inv_cur = p.inv_norm[(c.t_begin + tile_of(c, 0)) * 32 + j];
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): empty scoreboard at the loop head (see scan_kernel)
    __syncthreads();

    const int k_base = p.k / SETS, k_rem = p.k % SETS;
    const f32x16 zero16 = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t *const gb_row = p.gbound + (size_t)o.qg0 * PF_BOUND_CELLS + j;  // class j of the first owned query
    // a record of zeros nobody writes, one per workgroup: where the loads of a padded / zero query go (no branch
    // per query around the loads)
    const uint32_t *const idle_row = p.gbound_idle + (size_t)c.g * PF_BOUND_CELLS + j;

    // the derived bounds of the owned queries (one cell each, written by the queries' delegates)
    auto load_taus = [&](uint32_t (&tq)[RPO]) {
#pragma unroll
        for (int e = 0; e < RPO; ++e) {
            const bool live = (o.okmask >> e) & 1u;
            tq[e] = __hip_atomic_load(live ? gb_row - j + e * PF_BOUND_CELLS + PF_TAU_CELL : idle_row - j + PF_TAU_CELL,
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto take_taus = [&](const uint32_t (&tq)[RPO]) {
#pragma unroll
        for (int e = 0; e < RPO; ++e)
            if (tq[e] > tau[e]) {  // (a stale read can only be lower)
                tau[e] = tq[e];
                thr[e] = ord2f(tq[e]) - 2.f * PF_DELTA;
                if (j == 0) L.qthr[o.ql0 + e] = thr[e];
            }
    };
    auto any_without_bound = [&]() -> bool {
        bool none = false;
#pragma unroll
        for (int e = 0; e < RPO; ++e) none = none || (((o.okmask >> e) & 1u) && tau[e] == 0u);
        return __builtin_amdgcn_ballot_w64(none) != 0ull;
    };
    // delegate (wave 0): the class maxima of ONE query, half-wave h takes the sets h, h + 2
    constexpr int DS = SETS >= 2 ? SETS / 2 : 1;   // sorts per delegated query
    auto load_cells = [&](int dq, uint32_t (&cell)[DS]) {
        const uint32_t *rec = p.gbound + (size_t)((int)blockIdx.y * (32 * NQB) + dq) * PF_BOUND_CELLS;
#pragma unroll
        for (int i = 0; i < DS; ++i)
            cell[i] = __hip_atomic_load(rec + (SETS >= 2 ? 2 * i + h : 0) * 32 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto derive_cells = [&](int dq, const uint32_t (&cell)[DS]) {
        uint32_t t = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < DS; ++i) {
            const uint32_t sorted = sort32_desc_u32(cell[i], lane);
            const int sset = SETS >= 2 ? 2 * i + h : 0;
            const int ks = k_base + (sset < k_rem ? 1 : 0);
            const uint32_t kth = (uint32_t)__shfl((int)sorted, (lane & 32) | (ks - 1));
            t = kth < t ? kth : t;
        }
        if constexpr (SETS >= 2) {
            const uint32_t other = (uint32_t)__shfl((int)t, lane ^ 32);
            t = other < t ? other : t;
        }
        if (lane == 0 && t != 0u)
            (void)__hip_atomic_fetch_max(p.gbound + (size_t)((int)blockIdx.y * (32 * NQB) + dq) * PF_BOUND_CELLS + PF_TAU_CELL, t,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // the old way, for a wave that finds no published bound at its end: all class maxima of the owned queries
    auto derive_own = [&]() {
#pragma unroll
        for (int e = 0; e < RPO; ++e) {
            const bool live = (o.okmask >> e) & 1u;
            uint32_t t = 0xffffffffu;
#pragma unroll
            for (int s = 0; s < SETS; ++s) {
                const uint32_t v = __hip_atomic_load(live ? gb_row + e * PF_BOUND_CELLS + s * 32 : idle_row + s * 32,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t sorted = sort32_desc_u32(v, lane);
                const int ks = k_base + (s < k_rem ? 1 : 0);
                const uint32_t kth = (uint32_t)__shfl((int)sorted, (lane & 32) | (ks - 1));
                t = kth < t ? kth : t;
            }
            if (t > tau[e]) {
                tau[e] = t;
                thr[e] = ord2f(t) - 2.f * PF_DELTA;
                if (j == 0) L.qthr[o.ql0 + e] = thr[e];
            }
        }
    };

    // three cursors over the same sequence of publish points: publish after tile P, the delegates derive after tile
    // P + PF_DERIVE_LAG (the atomics of every workgroup have had two tile times to land and the lines are quiet
    // again), every wave reads after tile P + PF_READ_LAG
    const int PF_DERIVE_LAG = p.derive_lag, PF_READ_LAG = p.read_lag;   // (developer tuning: CRAG_PF_LAGS)
    int pub_at = pf_next_pub(-1, c.n_tiles, PF_READ_LAG);
    int der_base = pub_at, rd_base = pub_at;
    int next_read = rd_base + PF_READ_LAG;
    const int dq0 = c.g % (32 * NQB);   // (G >= 32 * NQB on a whole MI355X: one delegated query per workgroup)
    int buf = 0;
    for (int ti = 0; ti < c.n_tiles; ++ti) {
        const uint32_t vnext = voff(ti + 1);
        const int64_t tile = c.t_begin + tile_of(c, ti);
        const int64_t row = tile * 32 + j;
        const bool rd = ti == next_read;                                   // uniform
        const bool dv = (w == 0) && (ti == der_base + PF_DERIVE_LAG);      // uniform per wave
        // operands of this tile's epilogue, behind the B loads of this tile (in flight) and in front of the next one's
        float inv_nxt = 1.f;
        if constexpr (!MIRROR) inv_nxt = p.inv_norm[(c.t_begin + tile_of(c, ti + 1 < c.n_tiles ? ti + 1 : ti)) * 32 + j];
        uint32_t mword[RPO], tq[RPO], cell[DS];
#pragma unroll
        for (int e = 0; e < RPO; ++e)
            mword[e] = p.mask ? p.mask[(size_t)(((o.okmask >> e) & 1u) ? o.qg0 + e : 0) * (size_t)p.mask_stride_w + tile]
                              : 0xffffffffu;
        if (rd) load_taus(tq);
        if (dv) load_cells(dq0, cell);
        // The whole tile before the first MFMA: the 16 loads of the next tile then leave back to back, 16 KiB
        // contiguous per wave.  Waiting fragment by fragment re-issues them in dribs and drabs between 2047 other
        // waves' (measured on one box, 1M rows x 64: 643 us with this wait, 657 without).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

        f32x16 acc[NQB];
        static_for<0, 8>([&](auto T) {
            constexpr int t8 = decltype(T)::value;
            f16x8 bf;
            if constexpr (MIRROR) {
                bf = __builtin_bit_cast(f16x8, b[t8]);
            } else {  // two loads = dims {16 t8 + 4h + 0..3} and {16 t8 + 8 + 4h + 0..3} of row j
                const f32x4 lo = __builtin_bit_cast(f32x4, b[2 * t8]) * inv_cur;
                const f32x4 hi = __builtin_bit_cast(f32x4, b[2 * t8 + 1]) * inv_cur;
                const f16x2 p0 = __builtin_convertvector((f32x2{lo[0], lo[1]}), f16x2);
                const f16x2 p1 = __builtin_convertvector((f32x2{lo[2], lo[3]}), f16x2);
                const f16x2 p2 = __builtin_convertvector((f32x2{hi[0], hi[1]}), f16x2);
                const f16x2 p3 = __builtin_convertvector((f32x2{hi[2], hi[3]}), f16x2);
                bf = f16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
                acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[qb][t8], bf, t8 == 0 ? zero16 : acc[qb], 0, 0, AUX);
            if constexpr (MIRROR) {
                b[t8] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + t8 * 1024, 0, AUX);
            } else {
                b[2 * t8] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + b_soff<PS>(2 * t8), 0, AUX);
                b[2 * t8 + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + b_soff<PS>(2 * t8 + 1), 0, AUX);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int ow = 0; ow < SCAN_WAVES; ++ow) {  // split-K: owner wave ow gets registers [ow*RPO, ow*RPO+RPO)
            typename PfLds<NQB>::slab_t v;
#pragma unroll
            for (int e = 0; e < RPO; ++e) v[e] = acc[(ow * RPO + e) >> 4][(ow * RPO + e) & 15];
            L.slab[buf][ow][w][lane] = v;
        }
        // One barrier per tile.  It also carries the (uniform) decision to flush the staging buffer: every wave
        // posts what it saw after its own appends of the previous tile -- the wave whose append came last saw the
        // final count -- and all waves OR the same eight words behind the barrier.  [__syncthreads_or is a
        // workgroup reduction with three barriers of its own.]
        if (lane == 0) L.want_flush[buf][w] = L.n_stage > (uint32_t)PF_FLUSH_ABOVE ? 1u : 0u;
        __syncthreads();
        {
            const u32x4 f0 = *reinterpret_cast<const u32x4 *>(&L.want_flush[buf][0]);
            const u32x4 f1 = *reinterpret_cast<const u32x4 *>(&L.want_flush[buf][4]);
            if ((f0[0] | f0[1] | f0[2] | f0[3] | f1[0] | f1[1] | f1[2] | f1[3]) != 0u) pf_sift<NQB, false>(p, L, c.t_begin * 32);
        }
        float sc[RPO];
#pragma unroll
        for (int e = 0; e < RPO; ++e) sc[e] = 0.f;
#pragma unroll
        for (int ww = 0; ww < SCAN_WAVES; ++ww) {
            const typename PfLds<NQB>::slab_t v = L.slab[buf][w][ww][lane];
#pragma unroll
            for (int e = 0; e < RPO; ++e) sc[e] += v[e];
        }
        buf ^= 1;
        if (dv) {
            derive_cells(dq0, cell);
            for (int dq = dq0 + p.G; dq < 32 * NQB; dq += p.G) {   // (fewer workgroups than queries of a pass: partitioned GPU)
                uint32_t more[DS];
                load_cells(dq, more);
                derive_cells(dq, more);
            }
        }
        if (ti == der_base + PF_DERIVE_LAG) der_base = pf_next_pub(der_base, c.n_tiles, PF_READ_LAG);
        if constexpr (MIRROR) {   // emergency derivations asked for on the previous tile (see below): wave w takes the
                                  // flagged queries = w mod 8.  (fp32 rows: the block costs the scan's registers 8-22 spilled
                                  // dwords and 12 % of its rate at 100 000 rows; there a scan overlapped with another
                                  // search's kernels may end in the overflow fallback -- slower, same results.)
            uint32_t need[NQB];
#pragma unroll
            for (int i = 0; i < NQB; ++i) need[i] = __builtin_amdgcn_readfirstlane((int)L.need[i]);
            bool any = false;
#pragma unroll
            for (int i = 0; i < NQB; ++i) any = any || need[i] != 0u;
            if (any) {   // (uniform; never taken while the delegates keep up)
#pragma unroll
                for (int i = 0; i < NQB; ++i) {
                    uint32_t mine = need[i] & (0x01010101u << w);   // queries 32 i + w, + 8, + 16, + 24
                    while (mine) {
                        const int b = __builtin_ctz(mine);
                        mine &= mine - 1u;
                        uint32_t more[DS];
                        load_cells(32 * i + b, more);
                        derive_cells(32 * i + b, more);
                    }
                    if (lane == 0 && (need[i] & (0x01010101u << w))) atomicAnd(&L.need[i], ~(need[i] & (0x01010101u << w)));
                }
            }
        }
        if (rd) {
            take_taus(tq);
            // No bound yet for one of the queries: its delegates are not running yet (a scan that shares the GPU with
            // another search's kernels gets its workgroups a few at a time, and the delegates of a query are four
            // particular workgroups).  Without a bound the wave passes EVERY row of the coming tiles and the search ends
            // in the overflow fallback (measured: two scans of 1M rows started together, 1.9 ms per search instead of
            // 0.33).  So the wave asks ITS OWN workgroup for an emergency derivation (L.need: one bit per query of the
            // pass; the waves share the flagged queries on the next tile, below) and reads again on the tile after.
            if (any_without_bound()) {
                if constexpr (MIRROR) {
                    if (j == 0) {
#pragma unroll
                        for (int e = 0; e < RPO; ++e)
                            if (((o.okmask >> e) & 1u) && tau[e] == 0u) atomicOr(&L.need[(o.ql0 + e) >> 5], 1u << ((o.ql0 + e) & 31));
                    }
                }
                next_read = ti + 1;   // (fp32-row kernels: no emergency derivation, but still no tile without asking again)
            } else {
                rd_base = pf_next_pub(rd_base, c.n_tiles, PF_READ_LAG);
                while (rd_base + PF_READ_LAG <= ti) rd_base = pf_next_pub(rd_base, c.n_tiles, PF_READ_LAG);  // (terminates: rd_base grows)
                next_read = rd_base + PF_READ_LAG;
            }
        }
        const bool row_ok = (row >= c.r_begin) && (row < c.r_end) && (inv_cur > 0.f);
        const int set = __builtin_amdgcn_readfirstlane((int)(tile & (int64_t)(SETS - 1)));
        bool pass[RPO];
        uint32_t ord[RPO];
#pragma unroll
        for (int e = 0; e < RPO; ++e) {
            const bool ok = row_ok && ((o.okmask >> e) & 1u) && ((mword[e] >> j) & 1u) && (sc[e] == sc[e]);
            sc[e] = ok ? sc[e] : __uint_as_float(0x7fc00000u);
            ord[e] = ok ? f2ord(sc[e]) : 0u;
            pass[e] = sc[e] >= thr[e];
        }
        // the tile feeds ONE of the sets (a scalar): a scalar branch per set, RPO updates inside -- not RPO * SETS
        // compare-and-select pairs on every tile
#pragma unroll
        for (int s = 0; s < SETS; ++s)
            if (s == set) {
#pragma unroll
                for (int e = 0; e < RPO; ++e)
                    if (ord[e] > lmax[e][s]) {
                        lmax[e][s] = ord[e];
                        dirty |= 1u << (e * SETS + s);
                    }
            }
        if (ti < PF_STASH) {  // no bound can have arrived yet: keep the scores, decide at the end
#pragma unroll
            for (int t = 0; t < PF_STASH; ++t)
                if (ti == t) {
#pragma unroll
                    for (int e = 0; e < RPO; ++e) stash[t][e] = sc[e];
                }
        } else {
            pf_stage<NQB>(L, o, sc, pass, (uint32_t)(row - c.t_begin * 32), p.flags, p.seq);
        }
        if (ti == pub_at) {  // uniform
#pragma unroll
            for (int e = 0; e < RPO; ++e) {
                // After the first tile every class is empty and every lane would publish (524 000 atomics on 2048 cells
                // at once): there only the tile's best row(s) per query do -- the k best of the first 8192 rows are
                // their tiles' best with high probability, so the first bound is as good, from 16 000 atomics (8 per
                // class).  A workgroup's first tile feeds ONE of the sets: one sort per owned query.
                // (pub0 = SETS, or 8 when k_s nears 32: the bound then needs nearly EVERY class of the set to be
                // populated after this first exchange -- with SETS rows per workgroup 4 % of the queries of a k = 128
                // search found a class still empty, passed every row of the next tiles and pushed the whole search into
                // the fallback)
                uint32_t first_cut = 0u;
                if (ti == 0) {
                    uint32_t v0 = lmax[e][0];
#pragma unroll
                    for (int s = 1; s < SETS; ++s) v0 = (s == set) ? lmax[e][s] : v0;
                    if constexpr (SETS == 1) first_cut = half_max_u32(v0);
                    else first_cut = (uint32_t)__shfl((int)sort32_desc_u32(v0, lane), (lane & 32) | (p.pub0 - 1));
                }
#pragma unroll
                for (int s = 0; s < SETS; ++s) {
                    const uint32_t v = lmax[e][s];
                    // only a class maximum above the bound can lift the bound
                    const bool lift = ((dirty >> (e * SETS + s)) & 1u) && v > tau[e] && v >= first_cut;
                    if (lift)
                        (void)__hip_atomic_fetch_max(gb_row + e * PF_BOUND_CELLS + s * 32, v, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            dirty = 0u;
            pub_at = pf_next_pub(pub_at, c.n_tiles, PF_READ_LAG);
        }
        inv_cur = inv_nxt;
    }
    if (c.n_tiles > 0) {  // the stashed tiles, against the last bounds published
        {
            uint32_t tq[RPO];
            load_taus(tq);
            take_taus(tq);
            if (any_without_bound()) derive_own();
        }
        // One barrier: sift first if the staging buffer is nearly full (decided as in the loop: each wave posts what it
        // saw after its own last append), and learn whether EVERY wave holds bounds -- then the stashed tiles add a
        // handful of candidates and are staged back to back.  Without a bound a tile can add 2048 entries: a check
        // (a barrier) in front of each.
        const bool unbounded = any_without_bound();
        auto sift_if_full = [&](bool post_unbounded) -> bool {
            if (lane == 0) L.want_flush[buf][w] = (L.n_stage > (uint32_t)PF_FLUSH_ABOVE ? 1u : 0u) | (post_unbounded ? 2u : 0u);
            __syncthreads();
            const u32x4 f0 = *reinterpret_cast<const u32x4 *>(&L.want_flush[buf][0]);
            const u32x4 f1 = *reinterpret_cast<const u32x4 *>(&L.want_flush[buf][4]);
            buf ^= 1;
            const uint32_t f = f0[0] | f0[1] | f0[2] | f0[3] | f1[0] | f1[1] | f1[2] | f1[3];
            if ((f & 1u) != 0u) pf_sift<NQB, false>(p, L, c.t_begin * 32);
            return (f & 2u) != 0u;
        };
        const bool careful = sift_if_full(unbounded);   // (uniform over the workgroup)
#pragma unroll
        for (int t = 0; t < PF_STASH; ++t) {
            if (careful && t > 0) (void)sift_if_full(false);
            bool pass[RPO];
#pragma unroll
            for (int e = 0; e < RPO; ++e) pass[e] = stash[t][e] >= thr[e];
            // (a tile the workgroup does not have left NaN in the stash: nothing passes)
            pf_stage<NQB>(L, o, stash[t], pass, (uint32_t)(tile_of(c, t) * 32 + j), p.flags, p.seq);
        }
    }
    pf_sift<NQB, true>(p, L, c.t_begin * 32);
}

// ---- K3: candidates -> exact top-k.  One 256-thread workgroup per query ---------------------------------------
// 1. the k-th largest approximate score among the candidates is the k-th largest over ALL rows (every row of the
//    approximate top-k passed its threshold); rows below it by more than 2*delta cannot be in the exact top-k.
// 2. the survivors are rescored exactly: 8 lanes per row redo the scan kernels' arithmetic -- per 128-dim slice
//    the fmaf chain of v_mfma_f32_32x32x2_f32 in the kernels' k order, the 8 slice sums added in wave order,
//    x (1/||row|| * 1/||q||), clamp -- so scores and order are bit-identical to the fp32 scan.
// 3. rank by counting among the exact keys.
// Up to FIN_THREADS candidates (the usual case since the scan sifts its staged candidates, round 4) the k-th
// approximate score is found by counting ranks, one candidate per thread.  Beyond that a bit search on the orderable
// score, three bits per round, which stops FIN_SKIP_BITS above the bottom: the value it ends with is the k-th value
// with its low bits cleared -- a LOWER bound on it, at most 4096 ulps (2.4e-4 for a score in [0.5, 1), 3e-5 at 0.12)
// below, so the survivor window is that much wider than 2 delta and nothing is lost.  [Round 3 finished the search
// exactly by ranking the last bucket: three more barriers for one or two rows fewer to rescore.]
constexpr int FIN_SKIP_BITS = 12;
constexpr int FIN_THREADS = SCAN_THREADS;   // all eight waves of a selection block work (round 3: four of them)
constexpr int FIN_ROUND = 4096;             // candidates examined per round (all of them, for k <= 128 on the bench's corpora)
constexpr int FIN_BEST = FIN_ROUND + 128;   // exact keys kept in LDS (a round's survivors + the running top-k)

// One 128-dim slice of the exact dot product: the query slice comes from LDS (fragment order), the row's 32
// float4 are fetched first, all of them (32 independent 16-byte loads in flight, four per 64-byte piece of the
// row: latency is everything here), then the fmaf chain runs in the scan's k order:
// s = 0..15, then component, then lane half (k = 0, 1 of one 32x32x2 MFMA).
template <int PS>
__device__ __forceinline__ float exact_slice_dot(const f32x4 *qslice /* [16][2] float4 in LDS */,
                                                 const f32x4 *ctile /* tile base + slice */, int jrow) {
    f32x4 c0[16], c1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        // float4 kq = 2s (and 2s + 1) of the slice: piece kq >> PS of the row
        c0[s] = ctile[(((2 * s) >> PS) * 32 + jrow) * (1 << PS) + ((2 * s) & ((1 << PS) - 1))];
        c1[s] = ctile[(((2 * s) >> PS) * 32 + jrow) * (1 << PS) + ((2 * s) & ((1 << PS) - 1)) + 1];
    }
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 q0 = qslice[2 * s], q1 = qslice[2 * s + 1];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            acc = __builtin_fmaf(q0[cc], c0[s][cc], acc);
            acc = __builtin_fmaf(q1[cc], c1[s][cc], acc);
        }
    }
    return acc;
}

// LDS of the selection role; the fallback role's split-K slab (64 KiB) shares the same bytes
struct FinLds {
    uint64_t best[FIN_BEST];
    uint2 lcand[FIN_ROUND];       // the candidates of the common case (all of them fit): x = score, y = row
    uint32_t surv[FIN_ROUND];     // rows to rescore in the current round
    f32x4 qs[SCAN_WAVES][16][2];  // the query in fragment order: [slice][s][lane half]
    int64_t best_id[FIN_BEST];    // external id of best[i]'s row, fetched beside the row itself
    unsigned long long hist[64];  // exchange buffer of the workgroup reductions
    int s_nbest, s_nsurv, s_rescored;
    uint32_t s_kth;
};
union FinFbLds {
    FinLds f;
    float2 slab[2][SCAN_WAVES][8][64];
};

// ONE launch behind the prefilter scan, two kinds of workgroups (512 threads each):
//  * blocks [0, nq * R): selection, R blocks per query.  Waves 0-3 do the work described above, waves 4-7 end at
//    once.  R = 1 for k <= 32.  Larger k (the reference's own dense k is 50, the hybrid lane's 100): every block of a
//    query repeats the cheap part (k-th approximate score among the candidates), rescores the survivors whose list
//    position is r mod R exactly (131 rows x 4 KiB per query at k = 100: 30 us for one workgroup), writes its own
//    top-k to scratch, and the block that arrives LAST at the query's ticket gathers and ranks the R lists.
//  * blocks [nq * R, nq * R + fb_blocks): the FALLBACK for a search whose candidate list overflowed (thousands of rows within
//    2 delta of the k-th best: boilerplate chunks embed identically).  They end at once unless the overflow flag is
//    set; then each runs the exact fp32 scan of its row range (scan_body: self-contained, it normalises the raw
//    queries itself), and the workgroup that finishes LAST merges the partial lists of every query -- no workgroup
//    ever waits for another one, so nothing can deadlock; the selection blocks of such a search write nothing.
//    [Round 2 launched the fallback as a kernel of its own between scan and selection, gated by the flag: 256 heavy
//    workgroups created and torn down per search to read one word, 2.1-2.6 us + a kernel boundary.]
//  * Both roles leave the workspace clean: the selection block of query q zeroes the query's candidate count and
//    class maxima when it is done with them (the next search's scan finds zeros); the overflow word holds the
//    SEQUENCE NUMBER of the search that overflowed (never reset: the next search compares with its own number).
template <int S, int PS>
__global__ __launch_bounds__(SCAN_THREADS) void finalize_fb_kernel(FinParams p) {
    __shared__ FinFbLds U;
    __shared__ int s_last;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int R = p.rsplit;  // selection blocks per query (1, or 4 / 8 for large k: the exact rescoring is shared)
    if ((int)blockIdx.x >= p.nq * R) {
        if (__hip_atomic_load(p.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.seq) return;
        const int b = (int)blockIdx.x - p.nq * R;
        scan_body<S, PS>(p.scan, b % p.scan.G, b / p.scan.G, U.slab);
        // every wave drains its own list stores, then the barrier; one lane releases them and draws a ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t t = __hip_atomic_fetch_add(p.fb_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == (uint32_t)(p.fb_blocks - 1)) ? 1 : 0;
            if (s_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (!s_last || wave >= MERGE_THREADS / 64) return;   // (whole waves end: the barriers below count the rest)
        for (int q = 0; q < p.nq; ++q) {
            merge_partials_body(p.merge, q);
            __syncthreads();
        }
        if (threadIdx.x == 0) __hip_atomic_store(p.fb_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // developer probe: timestamps (100 MHz) of the phases of query 0's selection blocks
#define FIN_T(PH_)                                                                                               \
    do {                                                                                                         \
        if (p.trace && (int)blockIdx.x < R && threadIdx.x == 0) p.trace[((int)blockIdx.x % R) * 16 + (PH_)] = wall_clock64(); \
    } while (0)
    FIN_T(0);
    constexpr int NT = FIN_THREADS;
    auto &best = U.f.best;
    auto &lcand = U.f.lcand;
    auto &surv = U.f.surv;
    auto &qs = U.f.qs;
    auto &best_id = U.f.best_id;
    auto &hist = U.f.hist;
    int &s_nbest = U.f.s_nbest, &s_nsurv = U.f.s_nsurv, &s_rescored = U.f.s_rescored;
    uint32_t &s_kth = U.f.s_kth;
    const int q = (int)blockIdx.x / R, rpart = (int)blockIdx.x % R, tid = threadIdx.x;
    // Everything the kernel needs first, in flight together (one memory round trip instead of a chain of four:
    // flag -> count -> candidates -> ...): the overflow flag, the candidate count, the first NT candidates (read
    // before the count is known -- the list has `cap` >= NT slots; the usual few hundred candidates are all among
    // them), the query's norm and its fp32 fragments.
    const uint2 *gcand = p.cand + (size_t)q * p.cap;
    const bool overflow = p.flags[0] == p.seq;
    const uint32_t total = p.count[q];
    const uint2 first = gcand[tid];
    const float qinv = p.qinv[q];
    // thread t < 256 = (slice w, s, h) fetches its float4 of the raw query (prep_queries_kernel's layout)
    f32x4 qfrag = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 256)
        qfrag = reinterpret_cast<const f32x4 *>(p.a32)[((size_t)((q >> 5) * SCAN_WAVES + (tid >> 5)) * 16 + ((tid >> 1) & 15)) * 64 + (tid & 1) * 32 + (q & 31)];
    // Statistics (bench.py's byte accounting): a record per query that only this workgroup updates -- read here,
    // written back at the end with plain stores.  [They were three atomic adds on one shared record: 192 same-line
    // device-scope atomics per search, whose acknowledgements the kernel's end had to wait for.]  p.stats is the
    // block of PF_STAT_SLOTS / PF_STAT_WS records of this search's WORKSPACE (searches on different streams use different
    // workspaces and never share a record); a search of more queries than that folds them (counts may be lost;
    // results never depend on them).
    unsigned long long *const stat = p.stats ? p.stats + (size_t)(q % (PF_STAT_SLOTS / PF_STAT_WS)) * 3 : nullptr;
    unsigned long long stat_old[3] = {0ull, 0ull, 0ull};
    if (stat && tid == 0 && R == 1) {
        stat_old[0] = stat[0];
        stat_old[1] = stat[1];
        stat_old[2] = stat[2];
    }
    // the scan is over: leave the query's state zeroed for the next search on this workspace (the count only once
    // every thread of every block of the query has its copy: behind the first barrier below, or by the block that
    // finishes last when the query has several)
    if (tid < PF_BOUND_CELLS && rpart == 0) p.gbound[(size_t)q * PF_BOUND_CELLS + tid] = 0u;
    if (overflow) {  // (uniform over the grid) the fallback blocks of this launch answer this search
        if (tid == 0 && rpart == 0) p.count[q] = 0u;   // (the other blocks of the query use their copy for nothing)
        return;
    }
    const int k = p.k;
    const int C = (int)(total < (uint32_t)p.cap ? total : (uint32_t)p.cap);
    const bool in_lds = C <= FIN_ROUND;
    if (tid < C) lcand[tid] = first;
    if (in_lds)
        for (int e = tid + NT; e < C; e += NT) lcand[e] = gcand[e];
    if (tid < 256) qs[tid >> 5][(tid >> 1) & 15][tid & 1] = qfrag;
    if (tid == 0) {
        s_nbest = 0;
        s_rescored = 0;
        s_kth = 0u;
    }
    __syncthreads();
    if (tid == 0 && R == 1) p.count[q] = 0u;
    FIN_T(1);   // first round trip done, candidates in LDS

    // 1. (a lower bound on) the k-th largest approximate score among the candidates
    uint32_t thr_ord = 0u;
    if (C > k) {
        if (C <= 64) {  // a few dozen candidates (k <= 24): rank by counting in one wave, one barrier
            if (tid < C) {
                const uint32_t mine = lcand[tid].x;
                int rank = 0;
                for (int i = 0; i < C; ++i) {
                    const uint32_t o = lcand[i].x;
                    rank += (o > mine || (o == mine && i < tid)) ? 1 : 0;
                }
                if (rank == k - 1) s_kth = mine & ~((1u << FIN_SKIP_BITS) - 1u);   // (the same value the searches below end with)
            }
            __syncthreads();
        } else {
            // Radix-16 select on the orderable score with the digits on the LANES: lane L tests digit L & 15 against one of
            // 32 subsets of the list (4 per wave), i.e. counts its candidates >= ans | (digit << lo) -- a read, a compare
            // and an add per candidate, no dependent cross-lane step --; two v_permlane swaps add a wave's four subsets, the
            // eight waves' counts meet in LDS behind one barrier, and the largest digit still reached by k candidates is a
            // ballot away.  The candidates sit just above a common threshold, so their high bits agree: the search starts
            // below the highest bit in which any two of them differ and ends FIN_SKIP_BITS above the bottom: four rounds for
            // a typical top-100 search (14 bits).  [Radix 64 -- 63 thresholds per candidate -- is VALU-bound: 3-6.5 us.]
            // [What this replaced, each measured at ~350 candidates: three bits per round with per-thread digit tallies
            // and 24 64-bit shuffles per round 9-10 us; ranking by counting 12 us; one wave with ballots + s_bcnt1, with
            // shuffles or with DPP wave sums, one or three bits per round: 4.4-5 us every time -- with ONE wave on its
            // SIMD every dependent step costs its full latency, and a bit search is nothing but dependent steps.  A radix
            // select with an LDS histogram serialises: a thousand atomics on one bin per pass.]
            const int ln = tid & 63;
            uint32_t vmax = 0u, vmin = 0xffffffffu;
            if (in_lds) {
                for (int e = tid; e < C; e += NT) {
                    const uint32_t o = lcand[e].x;
                    vmax = o > vmax ? o : vmax;
                    vmin = o < vmin ? o : vmin;
                }
            } else {
                for (int e = tid; e < C; e += NT) {
                    const uint32_t o = gcand[e].x;
                    vmax = o > vmax ? o : vmax;
                    vmin = o < vmin ? o : vmin;
                }
            }
            vmax = wave_reduce_u32(vmax, ln, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            vmin = wave_reduce_u32(vmin, ln, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
            uint32_t *const cnts = reinterpret_cast<uint32_t *>(hist);   // [2][8 waves][64 lanes] -- surv is free until step 2
            uint32_t *const xch = surv;
            (void)cnts;
            if (ln == 0) {
                xch[wave] = vmax;
                xch[8 + wave] = vmin;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NT / 64; ++i) {
                const uint32_t a = xch[i], b = xch[8 + i];
                vmax = a > vmax ? a : vmax;
                vmin = b < vmin ? b : vmin;
            }
            const uint32_t diff = vmax ^ vmin;
            const int top0 = diff ? 32 - __builtin_clz(diff) : 0;      // bits [top0, 32) are common to all candidates
            const int top = top0 > FIN_SKIP_BITS ? top0 : FIN_SKIP_BITS;
            uint32_t ans = top >= 32 ? 0u : (vmax >> top) << top;
            int round = 0;
            const int digit_of = ln & 15, subset = (ln >> 4) + 4 * wave;   // 16 digits x 32 candidate subsets
            for (int bit = top; bit > FIN_SKIP_BITS; ++round) {
                const int nb = bit - FIN_SKIP_BITS >= 4 ? 4 : bit - FIN_SKIP_BITS, lo = bit - nb;   // this round: bits [lo, bit)
                const uint32_t t = ans | ((uint32_t)digit_of << lo);   // (digit 0 and digits >= 2^nb are not looked at)
                uint32_t c = 0u;
                if (in_lds) {
                    int e = subset;
                    for (; e + 3 * 32 < C; e += 4 * 32) {   // (four reads in flight)
                        const uint32_t x0 = lcand[e].x, x1 = lcand[e + 32].x, x2 = lcand[e + 64].x, x3 = lcand[e + 96].x;
                        c += (x0 >= t ? 1u : 0u) + (x1 >= t ? 1u : 0u) + (x2 >= t ? 1u : 0u) + (x3 >= t ? 1u : 0u);
                    }
                    for (; e < C; e += 32) c += lcand[e].x >= t ? 1u : 0u;
                } else {
                    for (int e = subset; e < C; e += 32) c += gcand[e].x >= t ? 1u : 0u;
                }
                {   // the wave's four subsets of a digit: rows 0..3 of 16 lanes
                    const auto r16 = __builtin_amdgcn_permlane16_swap(c, c, false, false);
                    c = r16[0] + r16[1];
                    const auto r32 = __builtin_amdgcn_permlane32_swap(c, c, false, false);
                    c = r32[0] + r32[1];
                }
                uint32_t *const slot = xch + 16 + (round & 1) * 128;   // [wave][digit]
                if (ln < 16) slot[wave * 16 + ln] = c;
                __syncthreads();
                uint32_t total = 0u;
#pragma unroll
                for (int wv = 0; wv < NT / 64; ++wv) total += slot[wv * 16 + digit_of];
                const unsigned long long reach =
                    __builtin_amdgcn_ballot_w64((int)total >= k && ln >= 1 && ln < (1 << nb) && ln < 16);
                const uint32_t digit = reach ? 63u - (uint32_t)__builtin_clzll(reach) : 0u;   // (the counts fall with the digit)
                ans |= digit << lo;
                bit = lo;
            }
            // ans = the k-th value with its low FIN_SKIP_BITS bits cleared (a lower bound on it)
            if (tid == 0) s_kth = ans;
            __syncthreads();
        }
        thr_ord = f2ord(ord2f(s_kth) - 2.f * PF_DELTA);
    }
    FIN_T(2);   // k-th approximate score known

    // 2. survivors -> exact scores, 8 lanes per row (64 rows per sweep), in rounds of FIN_ROUND candidates
    const int grp = tid >> 3, sub = tid & 7;  // lane `sub` = K slice (the scan's wave w)
    for (int r0 = 0; r0 < C; r0 += FIN_ROUND) {
        const int rn = (C - r0) < FIN_ROUND ? (C - r0) : FIN_ROUND;
        if (tid == 0) s_nsurv = 0;
        __syncthreads();
        for (int e = tid; e < rn; e += NT) {
            const uint2 ce = in_lds ? lcand[e] : gcand[r0 + e];
            // (several blocks per query: block r rescores the candidates whose list position is r mod R)
            if (ce.x >= thr_ord && (((r0 + e) & (R - 1)) == rpart)) surv[atomicAdd(&s_nsurv, 1)] = ce.y;
        }
        __syncthreads();
        const int ns = s_nsurv;
        for (int e0 = 0; e0 < ns; e0 += NT / 8) {
            const int e = e0 + grp;
            const bool live = e < ns;
            const uint32_t row = live ? surv[e] : 0u;
            float part = 0.f, inv_row = 0.f;
            int64_t rid = -1;
            if (live) {
                if (sub == 0) {  // in flight together with the row's own loads
                    inv_row = p.inv_norm[row];
                    rid = p.ids ? p.ids[row] : (int64_t)row;
                }
                const f32x4 *ctile = reinterpret_cast<const f32x4 *>(p.corpus + (size_t)(row >> 5) * TILE_FLOATS) + (size_t)sub * 32 * 32;
                part = exact_slice_dot<PS>(&qs[sub][0][0], ctile, (int)(row & 31u));
            }
            // the 8 slice sums in wave order 0..7 (the scan kernels' split-K reduction order)
            float d = __shfl(part, (tid & 63 & ~7) | 0);
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) d += __shfl(part, (tid & 63 & ~7) | ww);
            if (live && sub == 0) {
                const float scale = inv_row * qinv;
                float sc = d * scale;
                sc = __builtin_amdgcn_fmed3f(sc, -1.f, 1.f);
                if (scale > 0.f && sc == sc) {
                    const uint32_t u = __float_as_uint(sc);
                    const uint32_t ord = u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
                    const int slot = atomicAdd(&s_nbest, 1);
                    best[slot] = mk64(ord, ~row);
                    best_id[slot] = rid;
                }
            }
        }
        __syncthreads();
        if (tid == 0) s_rescored += ns;
        if (r0 + FIN_ROUND < C && s_nbest > k) {  // more rounds follow: keep only the running top-k
            const int B = s_nbest;
            uint64_t mine[(FIN_BEST + NT - 1) / NT];
            int64_t mine_id[(FIN_BEST + NT - 1) / NT];
            int rank[(FIN_BEST + NT - 1) / NT];
            int n_mine = 0;
            for (int e = tid; e < B; e += NT) {
                const uint64_t m = best[e];
                int rk = 0;
                for (int i = 0; i < B; ++i) rk += (best[i] > m) ? 1 : 0;
                mine[n_mine] = m;
                mine_id[n_mine] = best_id[e];
                rank[n_mine++] = rk;
            }
            __syncthreads();
            for (int e = 0; e < n_mine; ++e)
                if (rank[e] < k) {
                    best[rank[e]] = mine[e];
                    best_id[rank[e]] = mine_id[e];
                }
            if (tid == 0) s_nbest = k;
            __syncthreads();
        }
    }

    FIN_T(3);   // survivors rescored
    // 3. rank by counting among the exact keys
    int B = s_nbest;
    unsigned long long rescored_all = (unsigned long long)s_rescored;
    if (R > 1) {
        // this block's own best keys (at most k of them can be in the query's top-k; its share is rarely larger) ->
        // global scratch; the block of the query that finishes LAST gathers the R lists and ranks them.  Nobody waits
        // for anybody.
        uint64_t *const xk = p.xkeys + ((size_t)q * R + rpart) * k;
        int64_t *const xi = p.xids + ((size_t)q * R + rpart) * k;
        if (B <= k) {   // (uniform) the usual case: everything, unsorted
            for (int e = tid; e < B; e += NT) {
                xk[e] = best[e];
                xi[e] = best_id[e];
            }
        } else {
            for (int e = tid; e < B; e += NT) {
                const uint64_t mine = best[e];
                int rank = 0;
                for (int i = 0; i < B; ++i) rank += (best[i] > mine) ? 1 : 0;
                if (rank < k) {
                    xk[rank] = mine;
                    xi[rank] = best_id[e];
                }
            }
        }
        if (tid == 0) p.xcount[(size_t)q * R + rpart] = make_uint2((uint32_t)(B < k ? B : k), (uint32_t)s_rescored);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its own stores, then the barrier
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t t = __hip_atomic_fetch_add(p.xticket + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == (uint32_t)(R - 1)) ? 1 : 0;
            if (s_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        FIN_T(4);   // own list written, ticket drawn
        if (!s_last) return;
        // gather: the R counts and the first NT / R slots of every list are requested together (one round trip; a list
        // longer than that -- k > 64 with four lists -- is completed behind it)
        const int per = NT / R, rr_mine = tid / per, e_mine = tid % per;   // (R is 4 or 8: NT / R = 128 or 64 slots per list)
        const uint2 cr_mine = p.xcount[(size_t)q * R + rr_mine];
        uint64_t k_mine = 0ull;
        int64_t i_mine = -1;
        if (e_mine < k) {   // (within the list's k slots; stale beyond its count, masked below)
            k_mine = p.xkeys[((size_t)q * R + rr_mine) * k + e_mine];
            i_mine = p.xids[((size_t)q * R + rr_mine) * k + e_mine];
        }
        if (e_mine == 0) hist[48 + rr_mine] = ((unsigned long long)cr_mine.y << 32) | cr_mine.x;
        __syncthreads();
        int base = 0, base_mine = 0;
        rescored_all = 0ull;
        for (int rr = 0; rr < R; ++rr) {
            const unsigned long long c2 = hist[48 + rr];
            if (rr == rr_mine) base_mine = base;
            base += (int)(uint32_t)c2;
            rescored_all += c2 >> 32;
        }
        if (e_mine < (int)cr_mine.x) {
            best[base_mine + e_mine] = k_mine;
            best_id[base_mine + e_mine] = i_mine;
        }
        for (int e = e_mine + per; e < (int)cr_mine.x; e += per) {
            best[base_mine + e] = p.xkeys[((size_t)q * R + rr_mine) * k + e];
            best_id[base_mine + e] = p.xids[((size_t)q * R + rr_mine) * k + e];
        }
        B = base;
        if (tid == 0) {
            __hip_atomic_store(p.xticket + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.count[q] = 0u;
            if (stat) {
                stat_old[0] = stat[0];
                stat_old[1] = stat[1];
                stat_old[2] = stat[2];
            }
        }
        __syncthreads();
        FIN_T(5);   // the R lists gathered
    }
    const int count = B < k ? B : k;
    for (int e = tid; e < B; e += NT) {
        const uint64_t mine = best[e];
        int rank = 0;
        for (int i = 0; i < B; ++i) rank += (best[i] > mine) ? 1 : 0;
        if (rank < k) {
            p.out_scores[(size_t)q * k + rank] = ord2f((uint32_t)(mine >> 32));
            p.out_ids[(size_t)q * k + rank] = best_id[e];
        }
    }
    for (int r = count + tid; r < k; r += NT) {
        p.out_scores[(size_t)q * k + r] = __uint_as_float(0x7fc00000u);
        p.out_ids[(size_t)q * k + r] = -1;
    }
    if (tid == 0) {
        p.out_counts[q] = count;
        if (stat) {
            stat[0] = stat_old[0] + (unsigned long long)total;
            stat[1] = stat_old[1] + rescored_all;
            if (q == 0) stat[2] = stat_old[2] + 1ull;
        }
    }
    FIN_T(6);
    if (p.trace && (int)blockIdx.x < R && threadIdx.x == 0) {
        p.trace[rpart * 16 + 8] = (unsigned long long)C;
        p.trace[rpart * 16 + 9] = (unsigned long long)s_rescored;
    }
#undef FIN_T
}

// number of positions i in [0, n) whose id is not greater than its predecessor's (ids[-1] = prev)
__global__ __launch_bounds__(256) void check_ids_kernel(const int64_t *ids, int64_t n, int64_t prev,
                                                        unsigned long long *out_bad) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t before = i ? ids[i - 1] : prev;
        c += ids[i] <= before ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out_bad, c);
}

// ------------------------------------------------------------------------------------------
// launchers (called from crag_api.cpp through crag_kernels.h)
// ------------------------------------------------------------------------------------------
hipError_t launch_scan(const ScanParams &p, int q_blocks, hipStream_t st, const char **kernel_name) {
    dim3 grid(p.G, q_blocks), block(SCAN_THREADS);
    const int ks = p.k <= 32 ? 1 : (p.k <= 64 ? 2 : 4);
    const char *name;
#define CRAG_LAUNCH(GRID_, ...)                                       \
    do {                                                              \
        name = "crag::" #__VA_ARGS__;                                 \
        hipLaunchKernelGGL((__VA_ARGS__), GRID_, block, 0, st, p);    \
    } while (0)
    const bool big = p.piece_shift == PS_BIG;   // (the row layout of an index with the fp16 mirror)
#define CRAG_LAUNCH_PS(GRID_, K_, ...)                                          \
    do {                                                                        \
        if (big) CRAG_LAUNCH(GRID_, K_<__VA_ARGS__, PS_BIG>);                    \
        else CRAG_LAUNCH(GRID_, K_<__VA_ARGS__, PS_SMALL>);                      \
    } while (0)
    if (p.unpipelined) {  // A/B testing only
        if (ks == 1) CRAG_LAUNCH_PS(grid, scan_kernel, 1);
        else if (ks == 2) CRAG_LAUNCH_PS(grid, scan_kernel, 2);
        else CRAG_LAUNCH_PS(grid, scan_kernel, 4);
    } else if (p.wide) {  // 64 queries per pass (q_blocks is even): the matrix-pipe-bound kernel
        const dim3 g2(p.G, q_blocks / 2);
        if (ks == 1) CRAG_LAUNCH_PS(g2, scan_pipe2_kernel, 1, 2);
        else if (ks == 2) CRAG_LAUNCH_PS(g2, scan_pipe2_kernel, 2, 2);
        else CRAG_LAUNCH_PS(g2, scan_pipe2_kernel, 4, 2);
    } else if (ks == 1) {
        if (big) CRAG_LAUNCH(grid, scan_pipe_kernel<PS_BIG>);
        else CRAG_LAUNCH(grid, scan_pipe_kernel<PS_SMALL>);
    } else if (ks == 2) {
        CRAG_LAUNCH_PS(grid, scan_pipe2_kernel, 2, 1);
    } else {
        CRAG_LAUNCH_PS(grid, scan_pipe2_kernel, 4, 1);
    }
#undef CRAG_LAUNCH_PS
#undef CRAG_LAUNCH
    if (kernel_name) *kernel_name = name;
    return hipGetLastError();
}

hipError_t launch_prep_queries(const PrepParams &p, int nq_pad, hipStream_t st) {
    hipLaunchKernelGGL(prep_queries_kernel, dim3(nq_pad), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_prefilter(const PfParams &p, int nqb, int passes, hipStream_t st, const char **kernel_name) {
    const dim3 grid(p.G, passes), block(SCAN_THREADS);
    const char *name = "";
#define CRAG_LAUNCH(...)                                             \
    do {                                                              \
        name = "crag::" #__VA_ARGS__;                                 \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, 0, st, p);     \
    } while (0)
    const bool mirror = p.corpus16 != nullptr;
#define CRAG_PICK(NQB_, SETS_)                                                      \
    do {                                                                             \
        if (mirror && p.nt) CRAG_LAUNCH(prefilter_kernel<NQB_, SETS_, true, true>);   \
        else if (mirror) CRAG_LAUNCH(prefilter_kernel<NQB_, SETS_, true, false>);     \
        else CRAG_LAUNCH(prefilter_kernel<NQB_, SETS_, false, false>);                \
    } while (0)
    if (nqb == 2) {
        if (p.sets == 1) CRAG_PICK(2, 1);
        else if (p.sets == 2) CRAG_PICK(2, 2);
        else CRAG_PICK(2, 4);
    } else {
        if (p.sets == 1) CRAG_PICK(1, 1);
        else if (p.sets == 2) CRAG_PICK(1, 2);
        else CRAG_PICK(1, 4);
    }
#undef CRAG_PICK
#undef CRAG_LAUNCH
    if (kernel_name) *kernel_name = name;
    return hipGetLastError();
}

hipError_t launch_finalize(const FinParams &p, hipStream_t st) {
    const dim3 grid(p.nq * p.rsplit + p.fb_blocks), block(SCAN_THREADS);
    const int ks = (p.k + 31) / 32;  // list slots of the fallback scan
    if (p.scan.piece_shift == PS_BIG) {
        if (ks == 1) hipLaunchKernelGGL((finalize_fb_kernel<1, PS_BIG>), grid, block, 0, st, p);
        else if (ks == 2) hipLaunchKernelGGL((finalize_fb_kernel<2, PS_BIG>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((finalize_fb_kernel<4, PS_BIG>), grid, block, 0, st, p);
    } else {
        if (ks == 1) hipLaunchKernelGGL((finalize_fb_kernel<1, PS_SMALL>), grid, block, 0, st, p);
        else if (ks == 2) hipLaunchKernelGGL((finalize_fb_kernel<2, PS_SMALL>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((finalize_fb_kernel<4, PS_SMALL>), grid, block, 0, st, p);
    }
    return hipGetLastError();
}

hipError_t launch_merge_partials(const MergeParams &p, int nq, hipStream_t st) {
    hipLaunchKernelGGL(merge_partials_kernel, dim3(nq), dim3(MERGE_THREADS), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_merge_results(const XMergeParams &p, hipStream_t st) {
    hipLaunchKernelGGL(merge_results_kernel, dim3(p.nq), dim3(MERGE_THREADS), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_store_rows(const float *rows, int dim, int64_t pos, int64_t n, float *corpus,
                             float *inv_norm, uint32_t *irregular, _Float16 *mirror, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    // the row layout follows the mirror: PS_BIG with it, PS_SMALL without (see PS_SMALL / PS_BIG)
    if (mirror) hipLaunchKernelGGL(store_rows_kernel<PS_BIG>, dim3((unsigned)n), dim3(256), 0, st, rows, dim, pos, corpus,
                                   inv_norm, irregular, mirror);
    else hipLaunchKernelGGL(store_rows_kernel<PS_SMALL>, dim3((unsigned)n), dim3(256), 0, st, rows, dim, pos, corpus,
                       inv_norm, irregular, mirror);
    return hipGetLastError();
}

hipError_t launch_load_rows(const float *corpus, int dim, int64_t pos, int64_t n, float *rows, int piece_shift,
                            hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (piece_shift == PS_BIG) hipLaunchKernelGGL(load_rows_kernel<PS_BIG>, dim3((unsigned)n), dim3(256), 0, st, corpus, dim, pos, rows);
    else hipLaunchKernelGGL(load_rows_kernel<PS_SMALL>, dim3((unsigned)n), dim3(256), 0, st, corpus, dim, pos, rows);
    return hipGetLastError();
}

hipError_t launch_count_eligible(const float *inv_norm, int64_t n, const uint32_t *mask,
                                 unsigned long long *out, hipStream_t st) {
    hipLaunchKernelGGL(count_eligible_kernel, dim3(1024), dim3(256), 0, st, inv_norm, n, mask, out);
    return hipGetLastError();
}

hipError_t launch_check_ids(const int64_t *ids, int64_t n, int64_t prev, unsigned long long *out_bad,
                            hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(check_ids_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, ids, n,
                       prev, out_bad);
    return hipGetLastError();
}

hipError_t launch_fill_ids(int64_t *ids, int64_t pos, int64_t n, int64_t first, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, pos, n,
                       first);
    return hipGetLastError();
}

}  // namespace crag
