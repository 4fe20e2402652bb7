// crag_kernels.h — parameter blocks and launchers shared by crag_search.hip and crag_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crag_arch.h"

namespace crag {

constexpr int TILE_ROWS = 32;                 // corpus rows per tile (= MFMA N)
constexpr int DIM = 1024;                     // padded vector width
constexpr int TILE_FLOATS = TILE_ROWS * DIM;  // 128 KiB per tile
constexpr int SCAN_WAVES = 8;                 // split-K ways per workgroup
constexpr int SCAN_THREADS = SCAN_WAVES * 64;
constexpr int KSLICE = DIM / SCAN_WAVES;      // 128 dims per wave
constexpr int GB_CELLS = 32;                  // global-bound buckets per query (>= k of the pipelined kernel)

struct ScanParams {
    const float *corpus;      // tile32 layout
    const float *inv_norm;    // [cap_rows] 1/||row||, 0 = never eligible
    const float *queries;     // [nq, dim] row-major fp32, raw (only scan_kernel, the A/B baseline, reads them)
    const float *a32;         // the same queries in A-fragment order, zero padded (prep_queries_kernel)
    const float *qinv;        // [nq_pad] 1/||q|| (0: zero / non-finite query)
    const uint32_t *gate;     // nullable; when set and *gate == 0 the scan exits at once (fallback launch)
    int wide;                 // 1: the 64-queries-per-pass kernel (two query blocks per pass)
    const uint32_t *mask;     // nullable; 32 rows per word
    int64_t mask_stride_w;    // words between consecutive queries' masks (0 = shared)
    uint2 *partial;           // [q_blocks][G][32][k] keys
    uint32_t *gbound;         // [q_blocks*32][GB_CELLS] score buckets, zero between searches
    int64_t n_rows;
    int64_t cap_rows;
    int nq;
    int dim;
    int k;
    int G;
    int nb;                   // global-bound buckets in use: min(k, GB_CELLS)
    int pub_rank;             // rank (0-based) of the key a workgroup publishes: nb * (pub_rank + 1) >= k
    int reverse;              // walk each workgroup's range back to front (alternates per search)
    int unpipelined;          // 1: use the unpipelined kernel also for k <= 32 (A/B testing, CRAG_UNPIPELINED=1)
    int piece_shift;          // layout of the fp32 rows: crag_piece_shift(index has the fp16 mirror)
};

struct MergeParams {
    const uint2 *partial;
    const int64_t *ids;       // [cap_rows] row position -> external id (nullable)
    uint32_t *gbound;         // zeroed per query after use (nullable)
    int64_t id_base;          // used when ids == nullptr
    int64_t *out_ids;
    float *out_scores;
    int32_t *out_counts;
    int k;
    int G;
};

// ---- prefilter path (fp16 MFMA scan + exact rescoring), see crag_search.hip --------------------------------
// per-query bound record: 128 class maxima (4 sets x 32 row classes), padded to five 128-byte lines so that the
// records of two queries never share a line
constexpr int PF_BOUND_CELLS = 160;
constexpr int PF_STAT_WS = 8;                 // search workspaces per index (crag_api.hip MAX_WS): a block of records each
constexpr int PF_STAT_SLOTS = 2048;           // per-query statistics records (summed on the host when read)
constexpr int PF_MIN_ROWS_PER_GROUP = 128;    // below this many rows per workgroup the plain fp32 scan is used

struct PrepParams {
    const float *queries;     // [nq, dim] row-major fp32
    int nq, dim;
    float *qinv;              // [nq_pad]
    float *a32;               // [nq_pad/32][8 waves][16][64] float4: raw queries in fp32 A-fragment order
    _Float16 *a16;            // nullable; [nq_pad/32][8 waves][8][64][8]: unit queries in fp16 A-fragment order
};

struct PfParams {
    const float *corpus;
    const _Float16 *corpus16; // fp16 mirror of the unit rows in B-operand order (nullable: scan the fp32 rows)
    const float *inv_norm;
    const _Float16 *a16;
    const float *qinv;
    const uint32_t *mask;
    int64_t mask_stride_w;
    uint32_t *gbound;         // [nq_pad][PF_BOUND_CELLS] class maxima (orderable scores, atomic max)
    const uint32_t *gbound_idle;  // [>= G][PF_BOUND_CELLS] zeros that nothing writes: where the loads of dead queries go
    uint2 *cand;              // [nq_pad][cap]: x = orderable approximate score, y = row position
    uint32_t *count;          // [nq_pad]
    uint32_t *flags;          // [0] = sequence number of the search whose candidate list overflowed, [1] = fallback tickets
    uint32_t seq;             // this search's sequence number on its workspace (never 0)
    int64_t n_rows;
    int nq, k, G, reverse;
    int sets;                 // 1, 2 or 4 class sets (32 * sets >= k)
    int pub0;                 // rows of a workgroup's FIRST tile that publish their score per query (sets > 1): sets, or 8
    int nt;                   // corpus loads with the streaming cache policy (corpus larger than the Infinity Cache)
    int derive_lag, read_lag; // tiles between a publish of class maxima and the delegates' derivation / every wave's read
    int cap;
};

struct FinParams {
    const float *corpus;
    const float *inv_norm;
    const float *a32;
    const float *qinv;
    const uint2 *cand;
    uint32_t *count;            // [nq_pad]; zeroed per query after use
    uint32_t *gbound;           // [nq_pad][PF_BOUND_CELLS]; zeroed per query after use
    const uint32_t *flags;
    uint32_t seq;
    const int64_t *ids;
    int64_t *out_ids;
    float *out_scores;
    int32_t *out_counts;
    unsigned long long *stats;  // nullable; the workspace's PF_STAT_SLOTS / PF_STAT_WS records {candidates, rescored rows, searches}
    int k, cap;
    int nq;                     // selection blocks of the launch: [0, nq * rsplit)
    int rsplit;                 // selection blocks per query (R)
    uint64_t *xkeys;            // R > 1: [nq][R][k] exact keys of each block's own top-k
    int64_t *xids;              //        [nq][R][k] their external ids
    uint2 *xcount;              //        [nq][R] {keys written, rows rescored}
    uint32_t *xticket;          //        [nq] arrival tickets, zero between searches
    // fallback role (flags[0] != 0: a candidate list overflowed): fb_blocks = scan.G * ceil(nq / 32) workgroups behind
    // the selection blocks run the exact fp32 scan, the one that finishes last merges
    int fb_blocks;
    uint32_t *fb_done;          // ticket counter, zero between searches
    unsigned long long *trace;  // nullable (developer probe, CRAG_PHASE_TRACE=1): 100 MHz timestamps of the phases of the
                                // selection blocks of query 0: [rpart * 16 + phase]

    ScanParams scan;
    MergeParams merge;
};

struct XMergeParams {
    const int64_t *ids;
    const float *scores;
    const int32_t *counts;
    int64_t *out_ids;
    float *out_scores;
    int32_t *out_counts;
    int64_t stride_ids, stride_scores, stride_counts;  // elements between consecutive lists
    int n_lists;
    int nq;
    int k;
};

// kernel_name (nullable) receives the name of the kernel that was launched (a string literal)
hipError_t launch_scan(const ScanParams &p, int q_blocks, hipStream_t st, const char **kernel_name);
hipError_t launch_prep_queries(const PrepParams &p, int nq_pad, hipStream_t st);
// passes = number of 32*nqb-query passes (grid.y); nqb = 1 or 2
hipError_t launch_prefilter(const PfParams &p, int nqb, int passes, hipStream_t st, const char **kernel_name);
hipError_t launch_finalize(const FinParams &p, hipStream_t st);
hipError_t launch_merge_partials(const MergeParams &p, int nq, hipStream_t st);
hipError_t launch_merge_results(const XMergeParams &p, hipStream_t st);
hipError_t launch_store_rows(const float *rows, int dim, int64_t pos, int64_t n, float *corpus,
                             float *inv_norm, uint32_t *irregular, _Float16 *mirror, hipStream_t st);
hipError_t launch_load_rows(const float *corpus, int dim, int64_t pos, int64_t n, float *rows, int piece_shift,
                            hipStream_t st);
// fp32 row layout (pieces of 2^shift float4 per row and tile): 5 for an index with the fp16 mirror, 2 without
inline int crag_piece_shift(bool has_mirror) { return has_mirror ? 5 : 2; }
hipError_t launch_count_eligible(const float *inv_norm, int64_t n, const uint32_t *mask,
                                 unsigned long long *out, hipStream_t st);
hipError_t launch_check_ids(const int64_t *ids, int64_t n, int64_t prev, unsigned long long *out_bad,
                            hipStream_t st);
hipError_t launch_fill_ids(int64_t *ids, int64_t pos, int64_t n, int64_t first, hipStream_t st);

}  // namespace crag
