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
// scan kernel
// ------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(ScanParams p) {
    __shared__ float2 slab[2][SCAN_WAVES][8][64];  // 64 KiB: [buf][producer wave][reg pair][lane]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = blockIdx.x, qb = blockIdx.y;
    const int j = lane & 31, h = lane >> 5;

    // this workgroup's row range, balanced in units of 8 rows (one 128-B line per k-quad)
    const int64_t n8 = (p.n_rows + 7) >> 3;
    const int64_t r_begin = ((n8 * g) / p.G) << 3;
    int64_t r_end = ((n8 * (g + 1)) / p.G) << 3;
    if (r_end > p.n_rows) r_end = p.n_rows;
    const int64_t t_begin = r_begin >> 5;
    const int64_t t_end = (r_end > r_begin) ? ((r_end + 31) >> 5) : t_begin;
    const int n_tiles = (int)(t_end - t_begin);

    // A operand: this wave's K slice of the 32 normalised queries of block qb
    f32x4 a[16];
    {
        const f32x4 *qa =
            reinterpret_cast<const f32x4 *>(p.qtiles + (size_t)qb * TILE_FLOATS + w * (KSLICE * 32)) + lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = qa[s * 64];
        // drain here, with the compiler's own builtin, so that its vmcnt scoreboard is empty
        // before the streaming loop: otherwise the loop-head merge keeps a conservative
        // vmcnt(4) on the A registers in every iteration and drains the prefetch ring
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }

    // corpus window of this workgroup as a buffer resource: loads past the last tile, and the
    // lanes of rows that belong to a neighbouring workgroup, are dropped by the range check
    const float *wg_base = p.corpus + (size_t)t_begin * TILE_FLOATS;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(wg_base), 0, (int)((uint32_t)n_tiles * (uint32_t)(TILE_FLOATS * 4)), 0x00020000);
    const uint32_t OOB = 0x80000000u;
    const uint32_t lane_off = (uint32_t)(w * (KSLICE * 32 * 4) + lane * 16);

    auto tile_voff = [&](int ti) -> uint32_t {  // ti = tile index relative to t_begin
        const int64_t row = (t_begin + ti) * 32 + j;
        const bool valid = (ti < n_tiles) && (row >= r_begin) && (row < r_end);
        return valid ? (uint32_t)ti * (uint32_t)(TILE_FLOATS * 4) + lane_off : OOB;
    };

    u32x4 b[16];
    {
        const uint32_t v0 = tile_voff(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v0 + s * 1024, 0, 0);
    }

    HalfList<S> list[2];
    list[0].clear();
    list[1].clear();

    // owner bookkeeping: after the LDS reduction this wave holds accumulator registers
    // r = 2w, 2w+1 -> query (r&3) + 8*(r>>2) + 4*h of the block, corpus row j of the tile
    int qloc[2];
    bool qok[2];
    const uint32_t *mrow[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * w + e;
        qloc[e] = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int qglob = qb * 32 + qloc[e];
        qok[e] = qglob < p.nq;
        mrow[e] = p.mask ? p.mask + (size_t)(qok[e] ? qglob : 0) * (size_t)p.mask_stride_w : nullptr;
    }

    int buf = 0;
    for (int ti = 0; ti < n_tiles; ++ti) {
        const uint32_t vnext = tile_voff(ti + 1);
        const int64_t row = (t_begin + ti) * 32 + j;
        // epilogue operands, consumed ~4k cycles from now
        const float inv_cur = p.inv_norm[row];  // row < cap_rows: the tile exists
        uint32_t mword[2] = {0xffffffffu, 0xffffffffu};
        if (p.mask) {
            mword[0] = mrow[0][t_begin + ti];
            mword[1] = mrow[1][t_begin + ti];
        }

        // 16 steps of {4 MFMA on b[s]; refill b[s] from the next tile}.  The sched_barrier pins
        // that interleave: every load is issued 15 steps (one whole tile of MFMAs) before its
        // use, so 16 KiB per wave / 128 KiB per CU are always in flight.
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][0], __uint_as_float(b[s][0]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][1], __uint_as_float(b[s][1]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][2], __uint_as_float(b[s][2]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][3], __uint_as_float(b[s][3]), acc, 0, 0, 0);
            b[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vnext + s * 1024, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // split-K reduction through LDS (double-buffered: one barrier per tile)
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) slab[buf][w][pr][lane] = make_float2(acc[2 * pr], acc[2 * pr + 1]);
        __syncthreads();
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < SCAN_WAVES; ++ww) {  // fixed order: bit-reproducible
            const float2 v = slab[buf][ww][w][lane];
            d0 += v.x;
            d1 += v.y;
        }
        buf ^= 1;

        const bool row_ok = (row >= r_begin) && (row < r_end) && (inv_cur > 0.f);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float raw = (e == 0 ? d0 : d1);
            float sc = raw * inv_cur;
            sc = fminf(fmaxf(sc, -1.f), 1.f);  // pgvector clamps the similarity to [-1, 1]
            const bool ok = row_ok && qok[e] && ((mword[e] >> j) & 1u) && (raw == raw);
            const uint32_t khi = ok ? f2ord(sc) : 0u;
            const uint32_t klo = ok ? ~(uint32_t)row : 0u;
            list[e].insert(khi, klo, p.k, lane);
        }
    }

    // per-workgroup result: partial[qb][g][query][k]
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (!qok[e]) continue;
        uint2 *dst = p.partial + (((size_t)qb * p.G + g) * 32 + qloc[e]) * (size_t)p.k;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int pos = s * 32 + j;
            if (pos < p.k) dst[pos] = make_uint2(list[e].hi[s], list[e].lo[s]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// merge of the per-workgroup lists: one 256-thread workgroup per query
// ------------------------------------------------------------------------------------------
constexpr int MERGE_THREADS = 256;
constexpr int MERGE_CAP = 2048;       // candidates kept in LDS
constexpr int MERGE_RANK_MAX = 1024;  // above this many candidates: radix-select first
constexpr int CRAG_MAX_K_ = 128;      // = CRAG_MAX_K of include/crag_dense.h

__device__ __forceinline__ uint64_t key_of(uint2 v) { return mk64(v.x, v.y); }

__global__ __launch_bounds__(MERGE_THREADS) void merge_partials_kernel(MergeParams p) {
    __shared__ uint64_t cand[MERGE_CAP];
    __shared__ uint64_t win[CRAG_MAX_K_];
    __shared__ unsigned long long s_tau;
    __shared__ int s_cnt;
    __shared__ int hist[256];
    __shared__ int s_digit, s_need;

    const int q = blockIdx.x, tid = threadIdx.x;
    const int qb = q >> 5, ql = q & 31;
    const uint2 *base = p.partial + ((size_t)qb * p.G * 32 + ql) * (size_t)p.k;
    const size_t lstride = (size_t)32 * p.k;  // entries between consecutive workgroups' lists
    const int k = p.k, n_lists = p.G;
    const int total = n_lists * k;

    if (tid == 0) {
        s_tau = 0ull;
        s_cnt = 0;
    }
    __syncthreads();

    // 1. every final top-k key is >= the largest k-th key of any single list
    {
        unsigned long long m = 0ull;
        for (int l = tid; l < n_lists; l += MERGE_THREADS) {
            const unsigned long long kk = key_of(base[(size_t)l * lstride + (k - 1)]);
            m = kk > m ? kk : m;
        }
        if (m) atomicMax(&s_tau, m);
    }
    __syncthreads();
    const uint64_t tau = s_tau;

    // 2. compact the survivors into LDS
    for (int e = tid; e < total; e += MERGE_THREADS) {
        const int l = e / k, pos = e - l * k;
        const uint64_t kk = key_of(base[(size_t)l * lstride + pos]);
        if (kk != 0ull && kk >= tau) {
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
                    if (kk != 0ull && (kk & pmask) == prefix)
                        atomicAdd(&hist[(int)((kk >> shift) & 255ull)], 1);
                }
            }
            __syncthreads();
            // digit d with  sum_{d'>d} hist < need <= sum_{d'>=d} hist
            {
                int above = 0;
                for (int d = tid + 1; d < 256; ++d) above += hist[d];
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
        const int cnt = p.counts[(size_t)l * p.nq + q];
        if (pos < cnt) {
            const size_t src = ((size_t)l * p.nq + q) * k + pos;
            const int idx = atomicAdd(&s_cnt, 1);
            s_sc[idx] = f2ord(p.scores[src]);
            s_id[idx] = p.ids[src];
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

// queries [nq, dim] -> normalised, tile32 layout, zero-padded to a multiple of 32 queries.
// A zero / non-finite query becomes all-NaN so that none of its scores is eligible.
__global__ __launch_bounds__(256) void prep_queries_kernel(const float *queries, int nq, int dim,
                                                           float *qtiles) {
    __shared__ double sh[4];
    const int qi = blockIdx.x;  // padded slot
    const int kq = threadIdx.x;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (qi < nq) {
        const float *src = queries + (size_t)qi * dim;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * kq + c < dim) v[c] = src[4 * kq + c];
    }
    double ss = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    ss = block_sum_256(ss, sh);
    if (qi < nq) {
        const bool ok = (ss > 0.0) && (ss < 1.0e300) && (ss == ss);
        if (ok) {
            const double inv = 1.0 / sqrt(ss);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (float)((double)v[c] * inv);
        } else {
            const float nanv = __uint_as_float(0x7fc00000u);
            v = f32x4{nanv, nanv, nanv, nanv};
        }
    }
    float *dst = qtiles + (size_t)(qi >> 5) * TILE_FLOATS + ((size_t)kq * 32 + (qi & 31)) * 4;
    *reinterpret_cast<f32x4 *>(dst) = v;
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

// ------------------------------------------------------------------------------------------
// launchers (called from crag_api.cpp through crag_kernels.h)
// ------------------------------------------------------------------------------------------
hipError_t launch_scan(const ScanParams &p, int q_blocks, hipStream_t st) {
    dim3 grid(p.G, q_blocks), block(SCAN_THREADS);
    if (p.k <= 32)
        hipLaunchKernelGGL(scan_kernel<1>, grid, block, 0, st, p);
    else if (p.k <= 64)
        hipLaunchKernelGGL(scan_kernel<2>, grid, block, 0, st, p);
    else
        hipLaunchKernelGGL(scan_kernel<4>, grid, block, 0, st, p);
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

hipError_t launch_prep_queries(const float *queries, int nq, int dim, float *qtiles, hipStream_t st) {
    const int slots = ((nq + 31) / 32) * 32;
    hipLaunchKernelGGL(prep_queries_kernel, dim3(slots), dim3(256), 0, st, queries, nq, dim, qtiles);
    return hipGetLastError();
}

hipError_t launch_count_eligible(const float *inv_norm, int64_t n, const uint32_t *mask,
                                 unsigned long long *out, hipStream_t st) {
    hipLaunchKernelGGL(count_eligible_kernel, dim3(1024), dim3(256), 0, st, inv_norm, n, mask, out);
    return hipGetLastError();
}

hipError_t launch_fill_ids(int64_t *ids, int64_t pos, int64_t n, int64_t first, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, pos, n,
                       first);
    return hipGetLastError();
}

}  // namespace crag
