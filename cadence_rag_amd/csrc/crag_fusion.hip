// crag_fusion.hip — reciprocal-rank fusion of retrieval lanes on the GPU (BASELINE configs[4]:
// "dense top-100 fused with tech_tokens/BM25 lexical scores on GPU, batch = 64 queries").
// Semantics of the reference's _rrf_merge (/root/reference/app/retrieve.py:245-260): for every key
// score += 1/(k + rank) over the lanes in lane order (fp64, same summation order => bit-identical to
// the Python floats), the first row seen for a key is kept, result sorted by score descending with
// a STABLE sort, i.e. ties keep first-insertion order.
#include "crag_arch.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>

#include "../../include/crag_dense.h"

extern "C" void crag_set_error_(const char *msg);

namespace {

constexpr int FUSE_THREADS = 256;
constexpr int FUSE_MAX_ITEMS = 1024;
constexpr int FUSE_MAX_LANES = 8;

struct FuseParams {
    const int64_t *lane_ids[FUSE_MAX_LANES];    // [nq, width]
    const int32_t *lane_counts[FUSE_MAX_LANES]; // [nq]
    int width[FUSE_MAX_LANES];
    int n_lanes, nq, rrf_k, out_k;
    int64_t *out_ids;     // [nq, out_k]  -1 padded
    double *out_scores;   // [nq, out_k]  NaN padded
    uint32_t *out_lanes;  // [nq, out_k]  bit l set: lane l returned the key
    int32_t *out_counts;  // [nq]
};

// One workgroup per query.  Keys go into an LDS open-addressing table (slot -> first position of the key); the
// reciprocal-rank terms are added LANE BY LANE -- a lane's items in parallel, a barrier between lanes -- which is the
// reference's summation order for every key as long as no lane returns a key twice (none of the reference's lanes
// does: they are SELECTs over a primary key); a lane with a repeated key is detected while inserting and the query
// then takes the order-preserving quadratic path (round 2's only path: 61 us for 64 queries x 200 ids, all of them).
constexpr int FUSE_SLOTS = 2048;  // >= 2 * FUSE_MAX_ITEMS

// (developer build, -DCRAG_FUSE_TRACE: query 0's workgroup stamps the 100 MHz clock at its phase boundaries;
// scripts/probes/fuse_phase_trace.py)
#ifdef CRAG_FUSE_TRACE
__device__ unsigned long long g_fuse_trace[16];
#define FUSE_T(i)                                                                   \
    do {                                                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fuse_trace[i] = wall_clock64();  \
    } while (0)
#else
#define FUSE_T(i) ((void)0)
#endif

__device__ __forceinline__ uint32_t fuse_hash(int64_t k) {
    uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 40);
}

__global__ __launch_bounds__(FUSE_THREADS) void rrf_fuse_kernel(FuseParams p) {
    __shared__ int64_t key[FUSE_MAX_ITEMS];
    __shared__ alignas(16) double term[FUSE_MAX_ITEMS + 4];   // 1 / (k + rank) of the item; afterwards: the ranking keys
    __shared__ double score[FUSE_MAX_ITEMS];  // valid at first-occurrence positions
    __shared__ uint32_t lanes[FUSE_MAX_ITEMS];
    __shared__ uint8_t lane_of[FUSE_MAX_ITEMS];
    __shared__ int first[FUSE_MAX_ITEMS];     // first occurrence of the item's key
    __shared__ unsigned long long slot_key[FUSE_SLOTS];
    __shared__ int slot_first[FUSE_SLOTS];
    __shared__ uint32_t slot_lanes[FUSE_SLOTS];
    __shared__ int offs[FUSE_MAX_LANES + 1];
    __shared__ int s_dup, s_unique;
    const int q = blockIdx.x, tid = threadIdx.x;
    constexpr unsigned long long EMPTY = 0x8000000000000001ull;  // (not a valid id: ids are >= -1)
    FUSE_T(0);
    if (tid == 0) {
        int o = 0;
        for (int l = 0; l < p.n_lanes; ++l) {
            offs[l] = o;
            int c = p.lane_counts[l][q];
            c = c < 0 ? 0 : (c > p.width[l] ? p.width[l] : c);
            o += c;
        }
        offs[p.n_lanes] = o;
        s_dup = 0;
        s_unique = 0;
    }
    for (int i = tid; i < FUSE_SLOTS; i += FUSE_THREADS) {
        slot_key[i] = EMPTY;
        slot_first[i] = 0x7fffffff;
        slot_lanes[i] = 0u;
    }
    __syncthreads();
    FUSE_T(1);
    const int total = offs[p.n_lanes];
    for (int l = 0; l < p.n_lanes; ++l) {
        const int cnt = offs[l + 1] - offs[l];
        for (int r = tid; r < cnt; r += FUSE_THREADS) {
            const int i = offs[l] + r;
            const int64_t k = p.lane_ids[l][(size_t)q * p.width[l] + r];
            key[i] = k;
            term[i] = 1.0 / (double)(p.rrf_k + r + 1);
            lane_of[i] = (uint8_t)l;
            // find or claim the key's slot; remember the slot in first[] for now
            uint32_t sl = fuse_hash(k) & (FUSE_SLOTS - 1);
            bool placed = false;
            for (int tries = 0; tries < FUSE_SLOTS; ++tries) {   // (bounded: a full table must not spin -- the host keeps
                const unsigned long long prev = atomicCAS(&slot_key[sl], EMPTY, (unsigned long long)k);   // total <= SLOTS / 2)
                if (prev == EMPTY || prev == (unsigned long long)k) {
                    placed = true;
                    break;
                }
                sl = (sl + 1) & (FUSE_SLOTS - 1);
            }
            if (!placed) {   // cannot happen with total <= FUSE_MAX_ITEMS; if it ever does: the quadratic path answers
                s_dup = 1;
                continue;
            }
            first[i] = (int)sl;
            atomicMin(&slot_first[sl], i);
            if (atomicOr(&slot_lanes[sl], 1u << l) & (1u << l)) s_dup = 1;  // this lane already returned the key
        }
    }
    __syncthreads();
    FUSE_T(2);
    if (s_dup == 0) {
        // slot -> position of the key's first item; scores live at those positions
        for (int i = tid; i < total; i += FUSE_THREADS) {
            const int sl = first[i];
            const int f = slot_first[sl];
            if (f == i) {
                score[i] = 0.0;
                lanes[i] = slot_lanes[sl];
            }
            first[i] = f;
        }
        __syncthreads();
        for (int l = 0; l < p.n_lanes; ++l) {  // 0.0 + t(lane a) + t(lane b) + ...: the reference's order per key
            for (int i = offs[l] + tid; i < offs[l + 1]; i += FUSE_THREADS) score[first[i]] += term[i];
            __syncthreads();
        }
    } else {
        for (int i = tid; i < total; i += FUSE_THREADS) {
            const int64_t k = key[i];
            int f = i;
            for (int jj = 0; jj < i; ++jj)
                if (key[jj] == k) {
                    f = jj;
                    break;
                }
            first[i] = f;
        }
        __syncthreads();
        for (int i = tid; i < total; i += FUSE_THREADS) {
            if (first[i] != i) continue;
            double s = 0.0;
            uint32_t m = 0u;
            for (int jj = i; jj < total; ++jj)  // insertion order = lane order, then rank: the reference's order
                if (first[jj] == i) {
                    s += term[jj];
                    m |= 1u << lane_of[jj];
                }
            score[i] = s;
            lanes[i] = m;
        }
        __syncthreads();
    }
    FUSE_T(3);
    // rank = number of keys with a higher score, or the same score and an earlier first occurrence (the stable sort).
    // The scores are positive doubles, so their bit patterns order like the values: the ranking keys are the bit
    // patterns at first-occurrence positions and 0 everywhere else (never above, never equal), padded to a
    // multiple of 4 -- the loop reads four keys per step as two 16-byte broadcasts and compares integers.
    // (Round 3's loop read first[jj] and score[jj] per key and compared doubles: 19 of the kernel's 23 us.)
    unsigned long long *sb = reinterpret_cast<unsigned long long *>(term);
    const int total4 = (total + 3) & ~3;
    for (int i = tid; i < total4; i += FUSE_THREADS)
        sb[i] = (i < total && first[i] == i) ? (unsigned long long)__double_as_longlong(score[i]) : 0ull;
    __syncthreads();
    int mine_unique = 0;
    for (int i = tid; i < total; i += FUSE_THREADS) {
        if (first[i] != i) continue;
        ++mine_unique;
        const double s = score[i];
        const unsigned long long sv = sb[i];
        int rank = 0;
#pragma unroll 2
        for (int jj = 0; jj < total4; jj += 4) {
            const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&sb[jj]);
            const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(&sb[jj + 2]);
            rank += (a.x > sv || (a.x == sv && jj < i)) ? 1 : 0;
            rank += (a.y > sv || (a.y == sv && jj + 1 < i)) ? 1 : 0;
            rank += (b.x > sv || (b.x == sv && jj + 2 < i)) ? 1 : 0;
            rank += (b.y > sv || (b.y == sv && jj + 3 < i)) ? 1 : 0;
        }
        if (rank < p.out_k) {
            p.out_ids[(size_t)q * p.out_k + rank] = key[i];
            p.out_scores[(size_t)q * p.out_k + rank] = s;
            p.out_lanes[(size_t)q * p.out_k + rank] = lanes[i];
        }
    }
    if (mine_unique) atomicAdd(&s_unique, mine_unique);
    __syncthreads();
    FUSE_T(4);
    const int cnt = s_unique < p.out_k ? s_unique : p.out_k;
    if (tid == 0) p.out_counts[q] = cnt;
    for (int r = cnt + tid; r < p.out_k; r += FUSE_THREADS) {
        p.out_ids[(size_t)q * p.out_k + r] = -1;
        p.out_scores[(size_t)q * p.out_k + r] = __longlong_as_double(0x7ff8000000000000ll);
        p.out_lanes[(size_t)q * p.out_k + r] = 0u;
    }
    FUSE_T(5);
}

int ffail(const char *msg) {
    crag_set_error_(msg);
    return CRAG_EINVAL;
}

}  // namespace

#ifdef CRAG_FUSE_TRACE
extern "C" int crag_fuse_trace_read(unsigned long long *host16) {
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_fuse_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int crag_rrf_fuse(int n_lanes, const int64_t *const *d_lane_ids, const int32_t *const *d_lane_counts,
                             const int *lane_width, int nq, int rrf_k, int out_k, int64_t *d_out_ids,
                             double *d_out_scores, uint32_t *d_out_lanes, int32_t *d_out_counts, void *stream) {
    if (n_lanes <= 0 || n_lanes > FUSE_MAX_LANES) return ffail("rrf_fuse: n_lanes must be in [1, 8]");
    if (!d_lane_ids || !d_lane_counts || !lane_width || !d_out_ids || !d_out_scores || !d_out_lanes || !d_out_counts)
        return ffail("rrf_fuse: NULL pointer");
    if (nq < 0 || out_k <= 0 || rrf_k < 0) return ffail("rrf_fuse: bad sizes");
    if (nq == 0) return CRAG_OK;
    FuseParams p;
    int total = 0;
    for (int l = 0; l < n_lanes; ++l) {
        if (!d_lane_ids[l] || !d_lane_counts[l] || lane_width[l] < 0) return ffail("rrf_fuse: bad lane");
        p.lane_ids[l] = d_lane_ids[l];
        p.lane_counts[l] = d_lane_counts[l];
        p.width[l] = lane_width[l];
        total += lane_width[l];
    }
    if (total > FUSE_MAX_ITEMS) return ffail("rrf_fuse: more than 1024 items per query");
    p.n_lanes = n_lanes;
    p.nq = nq;
    p.rrf_k = rrf_k;
    p.out_k = out_k;
    p.out_ids = d_out_ids;
    p.out_scores = d_out_scores;
    p.out_lanes = d_out_lanes;
    p.out_counts = d_out_counts;
    hipLaunchKernelGGL(rrf_fuse_kernel, dim3((unsigned)nq), dim3(FUSE_THREADS), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[200];
        snprintf(buf, sizeof(buf), "rrf_fuse launch failed: %s", hipGetErrorString(e));
        crag_set_error_(buf);
        return CRAG_EHIP;
    }
    return CRAG_OK;
}

// ------------------------------------------------------------------------------------------------
// Exact-token lane (`tech_tokens && :tokens ... ORDER BY call_started_at DESC, id ASC LIMIT k`,
// /root/reference/app/retrieve.py:183-242) for a batch of queries.  Rows carry their technical tokens as
// 64-bit hashes in CSR form, STORED IN THE STATIC ORDER (call_started_at DESC, id ASC): CSR row r is the
// row at rank r, `order[r]` its position in the tables (for the row mask and the external id), so
// "ORDER BY ... LIMIT k" = the first k matching ranks and the scan streams through HBM coalesced.
//   kernel 1: the distinct query tokens go into an LDS hash table token -> 64-bit set of the queries that
//             contain it; one thread per rank looks its tokens up and ORs the sets; 64 ballots transpose
//             the wave's 64 x 64 bit matrix into one bitmap word per query (layout [word][query], so the
//             wave writes one contiguous line); the row mask is applied here
//   kernel 2: per query, walk the bitmap in chunks of 16384 ranks (popcount prefix) until k bits are found
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int TECH_MAX_Q = 64;
constexpr int TECH_MAX_QTOK = 32;
constexpr int TECH_SLOTS = 4096;  // >= 2 * TECH_MAX_Q * TECH_MAX_QTOK: load factor <= 0.5

struct TechParams {
    const int32_t *order;      // [n] table position of the row at rank r
    const int64_t *row_ptr;    // [n+1] CSR over ranks
    const uint64_t *tok;       // [nnz] token hashes
    const int64_t *ids;        // [n] external ids by table position (nullable: position)
    const uint64_t *qtok;      // [nq, TECH_MAX_QTOK]
    const int32_t *qtok_n;     // [nq]
    const uint32_t *mask;      // nullable row mask (bit per table POSITION), shared or per query
    int64_t mask_stride_w;
    uint64_t *bitmap;          // [words, nq]
    int64_t n, words;
    int nq, k;
    int64_t *out_ids;          // [nq, k]
    int32_t *out_counts;       // [nq]
};

__device__ __forceinline__ uint32_t tech_slot(uint64_t h) { return (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 52); }  // 12 bits

// 1024 threads per block: the 64 KiB token table allows two blocks per CU, and with 256-thread blocks that was 8 waves
// per CU -- two per SIMD for a kernel whose every step is a dependent load (row_ptr -> tokens -> LDS probe).
constexpr int TECH_MATCH_THREADS = 1024;
__global__ __launch_bounds__(TECH_MATCH_THREADS) void tech_match_kernel(TechParams p) {
    __shared__ unsigned long long s_key[TECH_SLOTS];   // 0 = empty (hash 0 is folded onto 1)
    __shared__ unsigned long long s_set[TECH_SLOTS];   // queries containing the token
    for (int i = threadIdx.x; i < TECH_SLOTS; i += blockDim.x) {
        s_key[i] = 0ull;
        s_set[i] = 0ull;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < p.nq * TECH_MAX_QTOK; i += blockDim.x) {
        const int q = i / TECH_MAX_QTOK, t = i % TECH_MAX_QTOK;
        if (t >= p.qtok_n[q]) continue;
        uint64_t h = p.qtok[i];
        h = h ? h : 1ull;
        uint32_t sl = tech_slot(h);
        for (;;) {
            const unsigned long long old = atomicCAS(&s_key[sl], 0ull, (unsigned long long)h);
            if (old == 0ull || old == h) {
                atomicOr(&s_set[sl], 1ull << q);
                break;
            }
            sl = (sl + 1) & (TECH_SLOTS - 1);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // whole waves stay together: every lane of a wave runs the same number of rounds
    for (int64_t r0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); r0 < p.words * 64; r0 += stride) {
        const int64_t r = r0 + lane;
        const bool live = r < p.n;
        uint64_t qset = 0ull;
        int64_t row = 0;
        if (live) {
            const int64_t t0 = p.row_ptr[r], t1 = p.row_ptr[r + 1];
            for (int64_t t = t0; t < t1; ++t) {
                uint64_t h = p.tok[t];
                h = h ? h : 1ull;
                uint32_t sl = tech_slot(h);
                for (;;) {
                    const unsigned long long kk = s_key[sl];
                    if (kk == h) {
                        qset |= s_set[sl];
                        break;
                    }
                    if (kk == 0ull) break;
                    sl = (sl + 1) & (TECH_SLOTS - 1);
                }
            }
            if (p.mask && qset) row = p.order[r];
        }
        unsigned long long mine = 0ull;
        if (__ballot(qset != 0ull) != 0ull) {  // wave-uniform: most 64-rank groups match no query at all
            // only the queries some row of the group matched get a ballot: the OR of the 64 sets, then a scalar walk
            // over its bits (a group of 64 rows typically matches a handful of the 64 queries: 64 ballots per group
            // were two thirds of this kernel's instructions)
            unsigned long long any = qset;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) any |= (unsigned long long)__shfl_xor((long long)any, o);
            uint32_t any_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)any);
            uint32_t any_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(any >> 32));
            for (int half = 0; half < 2; ++half) {
                uint32_t bits = half ? any_hi : any_lo;
                while (bits) {
                    const int q = half * 32 + __builtin_ctz(bits);
                    bits &= bits - 1u;
                    bool hit = (qset >> q) & 1ull;
                    if (hit && p.mask) hit = (p.mask[(size_t)q * (size_t)p.mask_stride_w + (row >> 5)] >> (row & 31)) & 1u;
                    const unsigned long long b = __ballot(hit);
                    if (lane == q) mine = b;
                }
            }
        }
        if (lane < p.nq) p.bitmap[(size_t)(r0 >> 6) * p.nq + lane] = mine;
    }
}

constexpr int TECH_CHUNK_WORDS = 256;  // words (of 64 ranks) per selection round: one per thread

__global__ __launch_bounds__(256) void tech_select_kernel(TechParams p) {
    __shared__ int s_cnt[256];
    __shared__ int s_base;
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t w0 = 0; w0 < p.words; w0 += TECH_CHUNK_WORDS) {
        const int base = s_base;
        if (base >= p.k) break;  // uniform: LIMIT k reached
        const int64_t w = w0 + tid;
        uint64_t bits = w < p.words ? p.bitmap[(size_t)w * p.nq + q] : 0ull;
        const int c = __popcll(bits);
        s_cnt[tid] = c;
        __syncthreads();
        int before = base;
        for (int i = 0; i < tid; ++i) before += s_cnt[i];  // 256 entries: fine
        int pos = before;
        while (bits && pos < p.k) {
            const int b = __builtin_ctzll(bits);
            bits &= bits - 1;
            const int64_t row = p.order[w * 64 + b];
            p.out_ids[(size_t)q * p.k + pos] = p.ids ? p.ids[row] : row;
            ++pos;
        }
        __syncthreads();
        if (tid == 255) s_base = before + c;
        __syncthreads();
    }
    if (tid == 0) {
        const int total = s_base;
        const int cnt = total < p.k ? total : p.k;
        p.out_counts[q] = cnt;
        for (int i = cnt; i < p.k; ++i) p.out_ids[(size_t)q * p.k + i] = -1;
    }
}

}  // namespace

extern "C" int crag_tech_lane(const int32_t *d_order, const int64_t *d_row_ptr, const uint64_t *d_tokens,
                              const int64_t *d_ids, int64_t n_rows, const uint64_t *d_query_tokens,
                              const int32_t *d_query_token_counts, int nq, int k, const uint8_t *d_row_mask,
                              int64_t mask_stride, uint64_t *d_bitmap_scratch, int64_t *d_out_ids,
                              int32_t *d_out_counts, void *stream) {
    if (!d_order || !d_row_ptr || !d_tokens || !d_query_tokens || !d_query_token_counts || !d_bitmap_scratch ||
        !d_out_ids || !d_out_counts)
        return ffail("tech_lane: NULL pointer");
    if (nq < 0 || nq > TECH_MAX_Q || k <= 0 || n_rows < 0) return ffail("tech_lane: need 0 <= nq <= 64, k > 0");
    if (nq == 0) return CRAG_OK;
    TechParams p;
    p.order = d_order;
    p.row_ptr = d_row_ptr;
    p.tok = d_tokens;
    p.ids = d_ids;
    p.qtok = d_query_tokens;
    p.qtok_n = d_query_token_counts;
    p.mask = (const uint32_t *)d_row_mask;
    p.mask_stride_w = mask_stride / 4;
    p.bitmap = d_bitmap_scratch;
    p.n = n_rows;
    p.words = (n_rows + 63) / 64;
    p.nq = nq;
    p.k = k;
    p.out_ids = d_out_ids;
    p.out_counts = d_out_counts;
    if (p.words == 0) p.words = 1;
    int64_t blocks = (p.words * 64 + TECH_MATCH_THREADS - 1) / TECH_MATCH_THREADS;
    if (blocks > 512) blocks = 512;  // persistent blocks (two per CU): the LDS token table is built once per block
    hipLaunchKernelGGL(tech_match_kernel, dim3((unsigned)blocks), dim3(TECH_MATCH_THREADS), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(tech_select_kernel, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[200];
        snprintf(buf, sizeof(buf), "tech_lane launch failed: %s", hipGetErrorString(e));
        crag_set_error_(buf);
        return CRAG_EHIP;
    }
    return CRAG_OK;
}

// ------------------------------------------------------------------------------------------------
// The lane from HOST token hashes: the upload slot (pinned host buffer + its device twin + the event of the last
// copy that read the host buffer) lives here, so that a call is: wait for the slot, pack, ONE async copy, two
// launches -- no tensor library in between (the Python path spent 65 us per call around 20 us of kernels).
// ------------------------------------------------------------------------------------------------
struct crag_upload_slot {
    void *host = nullptr;   // pinned: [64][32] uint64 query tokens, then [64] int32 counts
    void *dev = nullptr;
    hipEvent_t copied = nullptr;
    bool pending = false;
    int device = 0;
};
namespace {
constexpr size_t SLOT_QT_BYTES = (size_t)TECH_MAX_Q * 32 * sizeof(uint64_t);
constexpr size_t SLOT_BYTES = SLOT_QT_BYTES + (size_t)TECH_MAX_Q * sizeof(int32_t);
}  // namespace

extern "C" crag_upload_slot *crag_upload_slot_create(void) {
    crag_upload_slot *s = new (std::nothrow) crag_upload_slot();
    if (!s) {
        crag_set_error_("upload_slot: out of host memory");
        return nullptr;
    }
    if (hipGetDevice(&s->device) != hipSuccess || hipHostMalloc(&s->host, SLOT_BYTES, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(&s->dev, SLOT_BYTES) != hipSuccess ||
        hipEventCreateWithFlags(&s->copied, hipEventDisableTiming) != hipSuccess) {
        crag_set_error_("upload_slot: allocation failed");
        if (s->host) (void)hipHostFree(s->host);
        if (s->dev) (void)hipFree(s->dev);
        delete s;
        return nullptr;
    }
    memset(s->host, 0, SLOT_BYTES);
    return s;
}

extern "C" void crag_upload_slot_destroy(crag_upload_slot *s) {
    if (!s) return;
    if (s->pending) (void)hipEventSynchronize(s->copied);
    (void)hipEventDestroy(s->copied);
    (void)hipHostFree(s->host);
    (void)hipFree(s->dev);
    delete s;
}

extern "C" int crag_tech_lane_host(const int32_t *d_order, const int64_t *d_row_ptr, const uint64_t *d_tokens,
                                   const int64_t *d_ids, int64_t n_rows, const uint64_t *h_token_hashes,
                                   const int32_t *h_token_counts, int nq, int k, const uint8_t *d_row_mask,
                                   int64_t mask_stride, crag_upload_slot *slot, uint64_t *d_bitmap_scratch,
                                   int64_t *d_out_ids, int32_t *d_out_counts, void *stream) {
    if (!slot || !h_token_counts || (!h_token_hashes && nq > 0)) return ffail("tech_lane_host: NULL pointer");
    if (nq < 0 || nq > TECH_MAX_Q) return ffail("tech_lane_host: need 0 <= nq <= 64");
    if (nq == 0) return CRAG_OK;
    if (slot->pending) {   // the copy that last read the pinned buffer has left it
        if (hipEventSynchronize(slot->copied) != hipSuccess) return ffail("tech_lane_host: waiting for the upload slot failed");
        slot->pending = false;
    }
    uint64_t *qt = (uint64_t *)slot->host;
    int32_t *qn = (int32_t *)((char *)slot->host + SLOT_QT_BYTES);
    const uint64_t *src = h_token_hashes;
    for (int q = 0; q < nq; ++q) {   // distinct hashes, first occurrence kept (the SQL `&&` is a set overlap)
        const int n = h_token_counts[q];
        if (n < 0) return ffail("tech_lane_host: negative token count");
        uint64_t *row = qt + (size_t)q * 32;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const uint64_t h = src[i];
            bool seen = false;
            for (int j = 0; j < m; ++j) seen = seen || row[j] == h;
            if (seen) continue;
            if (m == 32) {
                crag_set_error_("tech_lane_host: more than 32 distinct tokens in a query (the caller splits it into passes)");
                return CRAG_E2BIG;
            }
            row[m++] = h;
        }
        qn[q] = m;
        src += n;
    }
    if (hipMemcpyAsync(slot->dev, slot->host, SLOT_QT_BYTES + (size_t)nq * sizeof(int32_t), hipMemcpyHostToDevice,
                       (hipStream_t)stream) != hipSuccess ||
        hipEventRecord(slot->copied, (hipStream_t)stream) != hipSuccess) {
        crag_set_error_("tech_lane_host: upload failed");
        return CRAG_EHIP;
    }
    slot->pending = true;
    return crag_tech_lane(d_order, d_row_ptr, d_tokens, d_ids, n_rows, (const uint64_t *)slot->dev,
                          (const int32_t *)((const char *)slot->dev + SLOT_QT_BYTES), nq, k, d_row_mask, mask_stride,
                          d_bitmap_scratch, d_out_ids, d_out_counts, stream);
}
