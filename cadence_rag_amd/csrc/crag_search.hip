// crag_search.hip — exact cosine top-k over a 1024-d fp32 corpus for gfx950 (MI355X, CDNA4).
//
// Replaces the pgvector exact-scan lane of the reference
// (/root/reference/app/retrieve.py:326-389: `ORDER BY embedding <=> q LIMIT k`).
// Written for CDNA4 only: 64-wide waves, v_mfma_f32_32x32x2_f32 (exact fp32), buffer loads
// with hardware range checking, ds_swizzle cross-lane exchange.  See DESIGN.md for the layout
// and the roofline arithmetic.
//
// HBM layout ("tile32"): the corpus is stored in tiles of 32 rows; inside a tile the float4
// holding dims [4*kq, 4*kq+3] of row j lives at float4 index kq*32 + j.  One wave-instruction
// `buffer_load_dwordx4` (64 lanes x 16 B) therefore reads 1 KiB of contiguous HBM AND lands
// exactly in the B-operand lane map of v_mfma_f32_32x32x2_f32 (lane l: row l&31, k-half l>>5).
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

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "crag_kernels.h"

namespace crag {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ ScanCtx make_ctx(const ScanParams &p) {
    ScanCtx c;
    c.lane = threadIdx.x & 63;
    c.w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    c.g = blockIdx.x;
    c.qb = blockIdx.y;
    c.j = c.lane & 31;
    c.h = c.lane >> 5;
    // this workgroup's row range, balanced in units of 8 rows (one 128-B line per k-quad)
    const int64_t n8 = (p.n_rows + 7) >> 3;
    c.r_begin = ((n8 * c.g) / p.G) << 3;
    c.r_end = ((n8 * (c.g + 1)) / p.G) << 3;
    if (c.r_end > p.n_rows) c.r_end = p.n_rows;
    c.t_begin = c.r_begin >> 5;
    const int64_t t_end = (c.r_end > c.r_begin) ? ((c.r_end + 31) >> 5) : c.t_begin;
    c.n_tiles = (int)(t_end - c.t_begin);
    c.reverse = p.reverse != 0;
    c.lane_off = (uint32_t)(c.w * (KSLICE * 32 * 4) + c.lane * 16);
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

__device__ __forceinline__ uint32_t tile_voff(const ScanCtx &c, int step) {
    const int ti = tile_of(c, step);
    const int64_t row = (c.t_begin + ti) * 32 + c.j;
    const bool valid = (step < c.n_tiles) && (row >= c.r_begin) && (row < c.r_end);
    return valid ? (uint32_t)ti * (uint32_t)(TILE_FLOATS * 4) + c.lane_off : 0x80000000u;
}

// A operand: this wave's K slice of the (up to) 32 queries of block qb, normalised in-kernel.
// Lane (i = lane&31, h = lane>>5) holds q[i][128w + 8s + 4h + 0..3] in a[s] — the same k
// permutation the tile32 corpus layout gives the B operand.  A zero / non-finite query becomes
// NaN so none of its scores is ever eligible.  `red` is 2 KiB of LDS scratch.
__device__ __forceinline__ void load_queries(const ScanParams &p, const ScanCtx &c, f32x4 (&a)[16],
                                             double *red) {
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
    double ss = 0.0;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) ss += (double)a[s][cc] * (double)a[s][cc];
    ss += __shfl_xor(ss, 32);
    if (c.h == 0) red[c.w * 32 + c.j] = ss;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int ww = 0; ww < SCAN_WAVES; ++ww) tot += red[ww * 32 + c.j];
    __syncthreads();
    const bool ok = (tot > 0.0) && (tot < 1.0e300) && (tot == tot);
    const double inv = ok ? 1.0 / sqrt(tot) : (double)__uint_as_float(0x7fc00000u);
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a[s][cc] = (float)((double)a[s][cc] * inv);
}

// score -> key for one owned (query, row) pair
__device__ __forceinline__ void make_key(float dot, float inv_row, bool ok, int64_t row, uint32_t &khi,
                                         uint32_t &klo) {
    float sc = dot * inv_row;
    sc = fminf(fmaxf(sc, -1.f), 1.f);  // pgvector clamps the similarity to [-1, 1]
    ok = ok && (dot == dot);
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
template <int S>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(ScanParams p) {
    __shared__ float2 slab[2][SCAN_WAVES][8][64];  // 64 KiB: [buf][producer wave][reg pair][lane]
    const ScanCtx c = make_ctx(p);
    const int lane = c.lane, w = c.w, j = c.j;

    const float *wg_base = p.corpus + (size_t)c.t_begin * TILE_FLOATS;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wg_base), 0, (int)((uint32_t)c.n_tiles * (uint32_t)(TILE_FLOATS * 4)), 0x00020000);

    u32x4 b[16];
    {
        const uint32_t v0 = tile_voff(c, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v0 + s * 1024, 0, 0);
    }
    f32x4 a[16];
    load_queries(p, c, a, reinterpret_cast<double *>(&slab[0][0][0][0]));
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
            b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + s * 1024, 0, 0);
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
            make_key(d[e], inv_cur, row_ok && c.qok[e] && ((mword[e] >> j) & 1u), row, khi, klo);
            list[e].insert(khi, klo, p.k, lane);
        }
    }
    write_lists<S>(p, c, list);
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

__global__ __launch_bounds__(SCAN_THREADS) void scan_pipe_kernel(ScanParams p) {
    __shared__ float2 slab[2][SCAN_WAVES][8][64];
    __shared__ double red[SCAN_WAVES][32];  // per-wave partial sums of squares of the 32 queries
    const ScanCtx c = make_ctx(p);
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
        for (int s = 0; s < 16; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v0 + s * 1024, 0, 0);
    }
    // A operand: raw queries, lane (i = lane&31, h = lane>>5) holds q[i][128w + 8s + 4h + 0..3]
    f32x4 a[16];
    {
        const int qi = c.qb * 32 + j;
        const bool have = qi < p.nq;
        const float *qrow = p.queries + (size_t)(have ? qi : 0) * p.dim;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int col = w * KSLICE + 8 * s + 4 * c.h;
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
        // four independent fp64 chains (one per component): a single dependent chain of 64 fp64 FMAs per
        // wave costs several microseconds of every launch
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) s4[cc] += (double)a[s][cc] * (double)a[s][cc];
        double ss = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        ss += __shfl_xor(ss, 32);
        if (c.h == 0) red[w][j] = ss;  // read after the first barrier of the tile loop
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
    float qinv[2] = {0.f, 0.f};  // 1/||q|| of the two owned queries of this half-wave

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
            if constexpr (cc == 3) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + s * 1024, 0, 0);
            if constexpr ((m & 1) == 0 && (m >> 1) < PIPE_OPS) {
                pipe_bg<(m >> 1)>(p, c, slab, wbuf ^ 1, prev, st);
            }
            if constexpr (m == 5) {
                // first tile only: the query norms (their partials were written before barrier 0)
                if (ti == 0) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        double tot = 0.0;
#pragma unroll
                        for (int ww = 0; ww < SCAN_WAVES; ++ww) tot += red[ww][c.qloc[e]];
                        const bool ok = c.qok[e] && (tot > 0.0) && (tot < 1.0e300) && (tot == tot);
                        qinv[e] = ok ? (float)(1.0 / sqrt(tot)) : 0.f;
                        if (!(qinv[e] < 3.0e38f)) qinv[e] = 0.f;
                    }
                }
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
    double red[SCAN_WAVES][NQB][32];               // partial sums of squares of the queries
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

template <int KS, int NQB>
__global__ __launch_bounds__(SCAN_THREADS) void scan_pipe2_kernel(ScanParams p) {
    __shared__ Pipe2Lds<KS, NQB> L;
    constexpr int RING = NQB == 2 ? 8 : 16;   // B prefetch ring depth (loads in flight per wave)
    constexpr int SLOTS = 64 * NQB;           // MFMAs per tile and wave
    const ScanCtx c = make_ctx(p);  // row range; its query fields are not used here
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
    for (int s = 0; s < RING; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vcur + s * 1024, 0, 0);

    // A operand: raw queries of the block(s), lane (i = lane&31, h) holds q[i][128w + 8s + 4h + 0..3];
    // 1/||q|| is applied to the score later
    f32x4 a[NQB][16];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int qi = ((int)blockIdx.y * NQB + qb) * 32 + j;
        const bool have = qi < p.nq;
        const float *qrow = p.queries + (size_t)(have ? qi : 0) * p.dim;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int col = w * KSLICE + 8 * s + 4 * c.h;
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
            a[qb][s] = v;
        }
        // after all loads are issued: four independent fp64 chains (one per component) instead of one
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) s4[cc] += (double)a[qb][s][cc] * (double)a[qb][s][cc];
        double ss = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        ss += __shfl_xor(ss, 32);
        if (c.h == 0) L.red[w][qb][j] = ss;  // combined after the first barrier of the tile loop
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int s = 0; s < KS; ++s) L.list[w][qb][e][s][lane] = make_uint2(0u, 0u);
            L.tinfo[w][qb][e][lane] = make_uint2(0u, 0u);  // "no previous tile": nothing eligible
            if ((lane & 31) == 0) {
                L.kth[w][qb][e][c.h] = make_uint2(0u, 0u);
                L.pub[w][qb][e][c.h] = 0u;
                L.qinv[w][qb][e][c.h] = 0.f;
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
                        if constexpr (base < 8) b[(base + ds) & 7] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vcur + (st2 + 8) * 1024, 0, 0);
                        else b[(base + ds) & 7] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + (st2 - 8) * 1024, 0, 0);
                    }
                }
            } else {
                constexpr int s = m >> 2, cc = m & 3;
                if constexpr (m == 0) acc[0] = CRAG_MFMA(a[0][s][cc], b[s][cc], zero16);
                else acc[0] = CRAG_MFMA(a[0][s][cc], b[s][cc], acc[0]);
                if constexpr (cc == 3) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + s * 1024, 0, 0);
            }
            if constexpr ((m & 1) == 0 && m >= OPS0) {
                constexpr int o = (m - OPS0) >> 1;
                if constexpr (o < PIPE_OPS) pipe2_bg<o, 0, NQB, KS>(p, c, c2, L, rbuf, st);
                else if constexpr (NQB == 2 && o < 2 * PIPE_OPS) pipe2_bg<o - PIPE_OPS, NQB - 1, NQB, KS>(p, c, c2, L, rbuf, st);
            }
            if constexpr (m == OPS0 + 3) {
                // first tile only: the query norms (partials were written before barrier 0)
                if (ti == 0) {
#pragma unroll
                    for (int q2 = 0; q2 < NQB; ++q2)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            double tot = 0.0;
#pragma unroll
                            for (int ww = 0; ww < SCAN_WAVES; ++ww) tot += L.red[ww][q2][c2.qloc[e]];
                            const bool ok = c2.qok[q2][e] && (tot > 0.0) && (tot < 1.0e300) && (tot == tot);
                            float qv = ok ? (float)(1.0 / sqrt(tot)) : 0.f;
                            if (!(qv < 3.0e38f)) qv = 0.f;
                            if ((lane & 31) == 0) L.qinv[w][q2][e][c.h] = qv;
                        }
                }
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

__global__ __launch_bounds__(MERGE_THREADS) void merge_partials_kernel(MergeParams p) {
    __shared__ uint64_t cand[MERGE_CAP];
    __shared__ uint64_t win[CRAG_MAX_K_];
    __shared__ __attribute__((aligned(16))) uint64_t heads[MERGE_HEADS + 8];
    __shared__ unsigned long long s_tau;
    __shared__ int s_cnt;
    __shared__ int hist[256];
    __shared__ int wave_tot[MERGE_THREADS / 64];
    __shared__ unsigned long long wave_max[MERGE_THREADS / 64];
    __shared__ int s_digit, s_need;

    const int q = blockIdx.x, tid = threadIdx.x;
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
        const int cnt = p.counts[(size_t)l * p.stride_counts + q];
        if (pos < cnt) {
            const size_t src = (size_t)q * k + pos;
            const int idx = atomicAdd(&s_cnt, 1);
            s_sc[idx] = f2ord(p.scores[(size_t)l * p.stride_scores + src]);
            s_id[idx] = p.ids[(size_t)l * p.stride_ids + src];
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
__global__ __launch_bounds__(256) void store_rows_kernel(const float *rows, int dim, int64_t pos,
                                                         float *corpus, float *inv_norm) {
    __shared__ double sh[4];
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
    float *dst = corpus + (size_t)(row >> 5) * TILE_FLOATS + ((size_t)kq * 32 + (row & 31)) * 4;
    *reinterpret_cast<f32x4 *>(dst) = v;
    double ss = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    ss = block_sum_256(ss, sh);
    if (kq == 0) {
        // zero or non-finite norm (NaN/Inf anywhere in the row) => never eligible
        const bool ok = (ss > 0.0) && (ss < 1.0e300) && (ss == ss);
        float inv = ok ? (float)(1.0 / sqrt(ss)) : 0.f;
        if (!(inv > 0.f) || !(inv < 3.0e38f)) inv = 0.f;
        inv_norm[row] = inv;
    }
}

__global__ __launch_bounds__(256) void load_rows_kernel(const float *corpus, int dim, int64_t pos,
                                                        float *rows) {
    const int64_t i = blockIdx.x;
    const int kq = threadIdx.x;
    const int64_t row = pos + i;
    const float *src = corpus + (size_t)(row >> 5) * TILE_FLOATS + ((size_t)kq * 32 + (row & 31)) * 4;
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
    if (p.unpipelined) {  // A/B testing only
        if (ks == 1) CRAG_LAUNCH(grid, scan_kernel<1>);
        else if (ks == 2) CRAG_LAUNCH(grid, scan_kernel<2>);
        else CRAG_LAUNCH(grid, scan_kernel<4>);
    } else if (p.wide) {  // 64 queries per pass (q_blocks is even): the matrix-pipe-bound kernel
        const dim3 g2(p.G, q_blocks / 2);
        if (ks == 1) CRAG_LAUNCH(g2, scan_pipe2_kernel<1, 2>);
        else if (ks == 2) CRAG_LAUNCH(g2, scan_pipe2_kernel<2, 2>);
        else CRAG_LAUNCH(g2, scan_pipe2_kernel<4, 2>);
    } else if (ks == 1) {
        CRAG_LAUNCH(grid, scan_pipe_kernel);
    } else if (ks == 2) {
        CRAG_LAUNCH(grid, scan_pipe2_kernel<2, 1>);
    } else {
        CRAG_LAUNCH(grid, scan_pipe2_kernel<4, 1>);
    }
#undef CRAG_LAUNCH
    if (kernel_name) *kernel_name = name;
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
                             float *inv_norm, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(store_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, rows, dim, pos, corpus,
                       inv_norm);
    return hipGetLastError();
}

hipError_t launch_load_rows(const float *corpus, int dim, int64_t pos, int64_t n, float *rows,
                            hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(load_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, corpus, dim, pos, rows);
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
