// crag_api.hip — host side of the C ABI declared in include/crag_dense.h.
// Owns the device corpus (tile32 layout, see crag_search.hip), the workspaces and the launch
// sequence prep_queries -> scan -> merge_partials.  No exceptions cross the ABI.

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/crag_dense.h"
#include "crag_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(CRAG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                        __FILE__, __LINE__);                                                   \
    } while (0)

bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

struct DevBuf {  // grow-only device scratch
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return CRAG_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(CRAG_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        bytes = want;
        return CRAG_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct EvSet {  // around one profiled search: start, before / after the scan kernel, end
    hipEvent_t e0, e1a, e1, e2, e3;  // e1a, e1: recorded back to back (what an event pair measures with nothing between)
};

}  // namespace

struct crag_index {
    int device = 0;
    int dim = 0;
    int64_t capacity = 0;   // rows requested
    int64_t cap_rows = 0;   // padded to a multiple of 32
    int64_t size = 0;
    int n_cu = 0;
    float *corpus = nullptr;
    _Float16 *corpus16 = nullptr;  // fp16 mirror of the unit rows for the prefilter scan (CRAG_NO_FP16_MIRROR=1: none)
    float *inv_norm = nullptr;
    int64_t *ids = nullptr;
    // search workspaces are per stream (up to MAX_WS streams): searches enqueued on different streams
    // may overlap on the GPU, same-stream searches are ordered by the stream itself
    static constexpr int MAX_WS = 8;   // (= crag::PF_STAT_WS; buffers are allocated on a workspace's first search)
    struct Workspace {
        hipStream_t stream = nullptr;
        bool in_use = false;
        DevBuf partial, gbound;
        // prepared queries (fragment order) and the prefilter path's per-query state
        DevBuf a32, a16, qinv, pf_gbound, pf_cand, pf_count, pf_flags, pf_xkeys, pf_xids, pf_xcount, pf_xticket;
        hipEvent_t done = nullptr;   // created with the index, recorded after every search that used this workspace
        uint32_t seq = 0;            // sequence number of the last prefilter search on this workspace (never 0 in use)
        bool done_recorded = false;  // ... once every workspace has an owner (until then nobody can take one over)
        // a search failed between its scan launch and its selection launch: the per-query state the selection kernel
        // leaves zeroed (class maxima, candidate counts, tickets) may hold the failed search's values -- the next search
        // on this workspace re-zeroes it first
        bool dirty = false;
        uint64_t last_use = 0;
    } ws[MAX_WS];
    // crag_index_search_pipelined: streams of the index's own, used in turn (3 by default; CRAG_PIPE_STREAMS=1..4)
    static constexpr int MAX_PIPE = 4;
    int n_pipe = 3;
    hipStream_t pipe[MAX_PIPE] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pipe_fork[MAX_PIPE] = {nullptr, nullptr, nullptr, nullptr}, pipe_done[MAX_PIPE] = {nullptr, nullptr, nullptr, nullptr};
    bool pipe_pending[MAX_PIPE] = {false, false, false, false};
    unsigned pipe_next = 0;
    uint64_t use_clock = 0;
    bool record_done = false;   // every workspace has an owner: from now on a search records its workspace's event
    int64_t last_id = INT64_MIN;  // largest id stored so far (ids are strictly ascending with the row position)
    // developer switches, read from the environment once, when the index is created
    bool env_no_wide = false, env_no_reverse = false, env_unpipelined = false, env_no_prefilter = false, env_no_rsplit = false;
    int env_pf_derive_lag = 2, env_pf_read_lag = 4;   // CRAG_PF_LAGS="d,r" (developer tuning; r <= 4 = the stashed tiles)
    int env_pf_nt = -1;                       // CRAG_PF_NT=0/1 forces the cache policy of the prefilter scan (developer switch)
    int64_t nt_above_bytes = 1536ll << 20;     // mirror bytes above which its loads stream (measured: no gain below ~1 GB)
    // a stored row whose norm lies outside [1e-30, 1e30]: the fp16 prefilter's error bound assumes normalised
    // rows in fp32's comfortable range, so such an index always takes the plain fp32 scan
    bool irregular = false;
    uint32_t *irregular_dev = nullptr;
    unsigned long long *pf_stats = nullptr;  // device: PF_STAT_SLOTS x {candidates, rescored rows, searches}
    unsigned long long *phase_trace = nullptr;  // device, 128 words; only with CRAG_PHASE_TRACE=1 (developer probe)
    const char *last_scan_kernel = "";  // name of the scan kernel the most recent search launched
    DevBuf stage_q, stage_rows, stage_ids, stage_mask, stage_out, scratch;
    std::mutex mu;
    int pass_parity = 0;  // alternate scan direction between searches (Infinity Cache reuse)
    int64_t env_fail_after_scan = 0;  // CRAG_TEST_FAIL_AFTER_SCAN=n (tests): the n-th prefilter search returns CRAG_EHIP
    int64_t pf_searches = 0;          // between its scan launch and its selection launch
    int profiling = 0;      // 0 = off, N = record HIP events around every N-th search
    int64_t prof_calls = 0;
    std::vector<EvSet> ev_pool;
    size_t ev_used = 0;
};

static_assert(crag_index::MAX_WS == crag::PF_STAT_WS, "one block of statistics records per workspace");

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int scan_groups(const crag_index *ix) {
    // one workgroup per CU; never more workgroups than 8-row groups
    int64_t n8 = (ix->size + 7) / 8;
    int64_t g = ix->n_cu;
    if (g > n8) g = n8;
    if (g < 1) g = 1;
    return (int)g;
}

int search_device(crag_index *ix, const float *d_queries, int nq, int k, const uint8_t *d_mask,
                  int64_t mask_stride, int64_t *d_out_ids, float *d_out_scores, int32_t *d_out_counts,
                  hipStream_t st) {
    if (nq <= 0) return CRAG_OK;
    int q_blocks = (nq + 31) / 32;
    // more than 32 queries: two query blocks share every corpus fragment (64 queries per pass)
    const bool wide = (nq > 32) && !ix->env_no_wide;
    if (wide) q_blocks = ((nq + 63) / 64) * 2;
    const int nq_pad = q_blocks * 32;
    const int G = scan_groups(ix);
    // the fp16 prefilter + exact rescoring path needs a few tiles per workgroup for its bounds to form;
    // small corpora take the plain fp32 scan (they are latency-, not bandwidth-bound anyway)
    const bool prefilter = !ix->env_no_prefilter && !ix->irregular &&
                           ix->size >= (int64_t)G * crag::PF_MIN_ROWS_PER_GROUP && (wide || nq <= 32);
    int rc;
    crag_index::Workspace *ws = nullptr;
    for (auto &w : ix->ws)
        if (w.in_use && w.stream == st) ws = &w;
    if (!ws)
        for (auto &w : ix->ws)
            if (!w.in_use) {
                w.in_use = true;
                w.stream = st;
                ws = &w;
                // the last free workspace: the next new stream takes one over, so from now on every search records
                // its workspace's completion event (until then: no event packet per search -- it cost a step 4-7 us
                // for every caller with two to four streams, and for crag_index_search_pipelined)
                if (&w == &ix->ws[crag_index::MAX_WS - 1]) ix->record_done = true;
                break;
            }
    if (!ws) {
        // every workspace belongs to some other stream: take the least recently used one and make this
        // stream wait for the last search that used it (its owner may even be gone by now)
        for (auto &w : ix->ws)
            if (!ws || w.last_use < ws->last_use) ws = &w;
        if (ws->done_recorded && !ws->dirty) {   // (a failed search recorded no event)
            HIP_TRY(hipStreamWaitEvent(st, ws->done, 0));
        } else {
            // its last search predates the moment the workspaces ran out (no event was recorded then): wait for the device
            HIP_TRY(hipDeviceSynchronize());
        }
        ix->record_done = true;
        ws->stream = st;
    }
    ws->last_use = ++ix->use_clock;
    if ((rc = ws->partial.ensure((size_t)q_blocks * G * 32 * (size_t)k * sizeof(uint2)))) return rc;
    {
        const size_t gb_bytes = (size_t)q_blocks * 32 * crag::GB_CELLS * sizeof(uint32_t);
        if (gb_bytes > ws->gbound.bytes) {  // (re)allocated buffers start zeroed; merge re-zeroes after use
            if ((rc = ws->gbound.ensure(gb_bytes))) return rc;
            HIP_TRY(hipMemsetAsync(ws->gbound.p, 0, ws->gbound.bytes, st));
        }
    }
    if ((rc = ws->a32.ensure((size_t)nq_pad * crag::DIM * sizeof(float)))) return rc;
    if ((rc = ws->qinv.ensure((size_t)nq_pad * sizeof(float)))) return rc;
    // candidates per query; a fuller list sends the search to the exact fallback.  k > 104 (k_s = 27 .. 32 of a set's
    // 32 class maxima: a weak bound) passes several thousand rows per query on a 1M-row corpus
    const int cap = (k > 104 && nq_pad <= 128) ? 32768 : 8192;
    // large k: several selection blocks per query share the exact rescoring (see finalize_fb_kernel)
    const int rsplit = (k <= 32 || ix->env_no_rsplit) ? 1 : (nq <= 16 ? 8 : (nq <= 128 ? 4 : 1));
    // Every allocation and memset of the search happens HERE, in front of its first launch: an allocation that fails
    // between the scan and the selection launch would leave the scan's per-query state behind (and a hipFree /
    // hipMalloc between two launches synchronises the device).
    if (prefilter) {
        if ((rc = ws->a16.ensure((size_t)nq_pad * crag::DIM * 2))) return rc;
        {   // n_cu idle records of zeros that nothing ever writes (zeroed once, when the buffer is allocated; at the
            // FRONT, so that no later search with fewer queries finds an old query record there), then the queries'
            // bound records (left zeroed by the selection kernel of every search)
            const size_t need = (size_t)(ix->n_cu + nq_pad) * crag::PF_BOUND_CELLS * sizeof(uint32_t);
            if (need > ws->pf_gbound.bytes) {
                if ((rc = ws->pf_gbound.ensure(need))) return rc;
                HIP_TRY(hipMemsetAsync(ws->pf_gbound.p, 0, ws->pf_gbound.bytes, st));
            }
        }
        if ((rc = ws->pf_cand.ensure((size_t)nq_pad * cap * sizeof(uint2)))) return rc;
        if (rsplit > 1) {   // scratch of the selection blocks that share a query (k > 32)
            const size_t slots = (size_t)nq_pad * 8;
            if ((rc = ws->pf_xkeys.ensure(slots * CRAG_MAX_K * sizeof(uint64_t)))) return rc;
            if ((rc = ws->pf_xids.ensure(slots * CRAG_MAX_K * sizeof(int64_t)))) return rc;
            if ((rc = ws->pf_xcount.ensure(slots * sizeof(uint2)))) return rc;
            const size_t tneed = (size_t)nq_pad * sizeof(uint32_t);
            if (tneed > ws->pf_xticket.bytes) {
                if ((rc = ws->pf_xticket.ensure(tneed))) return rc;
                HIP_TRY(hipMemsetAsync(ws->pf_xticket.p, 0, ws->pf_xticket.bytes, st));
            }
        }
        {   // per-query candidate counts and the overflow / ticket words: zero when allocated, kept clean by the kernels
            const size_t need = (size_t)nq_pad * sizeof(uint32_t);
            if (need > ws->pf_count.bytes) {
                if ((rc = ws->pf_count.ensure(need))) return rc;
                HIP_TRY(hipMemsetAsync(ws->pf_count.p, 0, ws->pf_count.bytes, st));
            }
            if (ws->pf_flags.bytes == 0) {
                if ((rc = ws->pf_flags.ensure(4 * sizeof(uint32_t)))) return rc;
                HIP_TRY(hipMemsetAsync(ws->pf_flags.p, 0, ws->pf_flags.bytes, st));
            }
        }
        if (ws->dirty) {  // the last search on this workspace died between scan and selection: nothing cleaned up
            HIP_TRY(hipMemsetAsync(ws->pf_gbound.p, 0, ws->pf_gbound.bytes, st));
            HIP_TRY(hipMemsetAsync(ws->pf_count.p, 0, ws->pf_count.bytes, st));
            HIP_TRY(hipMemsetAsync(ws->pf_flags.p, 0, ws->pf_flags.bytes, st));
            if (ws->pf_xticket.p) HIP_TRY(hipMemsetAsync(ws->pf_xticket.p, 0, ws->pf_xticket.bytes, st));
            ws->dirty = false;
        }
    }

    // per-workgroup corpus window must stay below the buffer-descriptor / OOB-marker limit
    const int64_t rows_per_g = (ix->size + G - 1) / G + 64;
    if (rows_per_g * (int64_t)(crag::DIM * 4) >= (int64_t)0x7ff00000)
        return fail(CRAG_EINVAL, "index too large for one device scan window (%lld rows)",
                    (long long)ix->size);

    EvSet *ev = nullptr;
    // (the sampled search is the one in the MIDDLE of every window of N: with a caller that synchronises every N searches
    // the first of a window starts on an idle GPU and is not the typical one)
    if (ix->profiling > 0 && (ix->prof_calls++ % ix->profiling) == ix->profiling / 2) {
        if (ix->ev_used == ix->ev_pool.size()) {
            EvSet t;
            HIP_TRY(hipEventCreate(&t.e0));
            HIP_TRY(hipEventCreate(&t.e1a));
            HIP_TRY(hipEventCreate(&t.e1));
            HIP_TRY(hipEventCreate(&t.e2));
            HIP_TRY(hipEventCreate(&t.e3));
            ix->ev_pool.push_back(t);
        }
        ev = &ix->ev_pool[ix->ev_used++];
        HIP_TRY(hipEventRecord(ev->e0, st));
    }

    // K0: 1/||q||, the queries in A-fragment order, reset of the prefilter state
    crag::PrepParams pp;
    pp.queries = d_queries;
    pp.nq = nq;
    pp.dim = ix->dim;
    pp.qinv = (float *)ws->qinv.p;
    pp.a32 = (float *)ws->a32.p;
    pp.a16 = prefilter ? (_Float16 *)ws->a16.p : nullptr;
    HIP_TRY(crag::launch_prep_queries(pp, nq_pad, st));

    crag::ScanParams sp;
    sp.wide = wide ? 1 : 0;
    sp.corpus = ix->corpus;
    sp.inv_norm = ix->inv_norm;
    sp.queries = d_queries;
    sp.a32 = (const float *)ws->a32.p;
    sp.qinv = (const float *)ws->qinv.p;
    sp.gate = nullptr;
    sp.dim = ix->dim;
    sp.mask = (const uint32_t *)d_mask;
    sp.mask_stride_w = mask_stride / 4;
    sp.partial = (uint2 *)ws->partial.p;
    sp.gbound = (uint32_t *)ws->gbound.p;
    sp.n_rows = ix->size;
    sp.cap_rows = ix->cap_rows;
    sp.nq = nq;
    sp.k = k;
    sp.G = G;
    sp.nb = k < crag::GB_CELLS ? k : crag::GB_CELLS;
    sp.pub_rank = (k + sp.nb - 1) / sp.nb - 1;
    sp.reverse = ix->pass_parity;
    ix->pass_parity ^= 1;
    if (ix->env_no_reverse) sp.reverse = 0;
    sp.unpipelined = (ix->env_unpipelined && !prefilter) ? 1 : 0;
    sp.piece_shift = crag::crag_piece_shift(ix->corpus16 != nullptr);

    crag::MergeParams mp;
    mp.partial = (const uint2 *)ws->partial.p;
    mp.ids = ix->ids;
    mp.gbound = (uint32_t *)ws->gbound.p;
    mp.id_base = 0;
    mp.out_ids = d_out_ids;
    mp.out_scores = d_out_scores;
    mp.out_counts = d_out_counts;
    mp.k = k;
    mp.G = G;

    if (ev) {
        HIP_TRY(hipEventRecord(ev->e1a, st));
        HIP_TRY(hipEventRecord(ev->e1, st));
    }
    if (prefilter) {
        // K1: fp16 scan -> candidates;  K2: exact rescoring + selection, and -- workgroups of the same launch that end
        // at once unless a candidate list overflowed -- the fp32 fallback scan + merge
        crag::PfParams fp;
        fp.corpus = ix->corpus;
        fp.corpus16 = ix->corpus16;
        fp.inv_norm = ix->inv_norm;
        fp.a16 = (const _Float16 *)ws->a16.p;
        fp.qinv = (const float *)ws->qinv.p;
        fp.mask = (const uint32_t *)d_mask;
        fp.mask_stride_w = mask_stride / 4;
        fp.gbound_idle = (const uint32_t *)ws->pf_gbound.p;
        fp.gbound = (uint32_t *)ws->pf_gbound.p + (size_t)ix->n_cu * crag::PF_BOUND_CELLS;
        fp.cand = (uint2 *)ws->pf_cand.p;
        fp.count = (uint32_t *)ws->pf_count.p;
        fp.flags = (uint32_t *)ws->pf_flags.p;
        if (++ws->seq == 0u) ws->seq = 1u;
        fp.seq = ws->seq;
        fp.n_rows = ix->size;
        fp.nq = nq;
        fp.k = k;
        fp.G = G;
        fp.reverse = sp.reverse;
        // class sets (32 * sets >= k; the bound is the k_s-th largest of a set's 32 class maxima, k_s = k / sets).  Two
        // sets up to k = 56 (k_s <= 28): measured at k = 50 against four sets (k_s = 12-13), 100 000 x 64: 78.9 vs
        // 82.3 us per step -- the half-wave sorts of every exchange cost more than the tighter bound returns
        fp.sets = k <= 24 ? 1 : (k <= 56 ? 2 : 4);
        fp.pub0 = (k + fp.sets - 1) / fp.sets >= 27 ? 8 : fp.sets;
        fp.cap = cap;
        // exchange lags in tiles: publish -> delegates derive -> every wave reads.  Mirror scan 2 / 4; the scan of the
        // fp32 rows (twice the time per tile, two stashed tiles) 1 / 2
        fp.derive_lag = ix->corpus16 ? ix->env_pf_derive_lag : 1;
        fp.read_lag = ix->corpus16 ? ix->env_pf_read_lag : 2;
        {   // streaming cache policy for a mirror far larger than the Infinity Cache (see prefilter_kernel)
            const int64_t streamed = ix->size * (int64_t)crag::DIM * 2;
            fp.nt = !ix->corpus16 ? 0 : (ix->env_pf_nt >= 0 ? ix->env_pf_nt : (streamed > ix->nt_above_bytes ? 1 : 0));
        }
        const int nqb = wide ? 2 : 1;
        ws->dirty = true;   // until the selection launch is in the stream
        HIP_TRY(crag::launch_prefilter(fp, nqb, nq_pad / (32 * nqb), st, &ix->last_scan_kernel));
        if (ix->env_fail_after_scan > 0 && ++ix->pf_searches == ix->env_fail_after_scan)
            return fail(CRAG_EHIP, "injected failure behind the scan launch (CRAG_TEST_FAIL_AFTER_SCAN)");
        if (ev) HIP_TRY(hipEventRecord(ev->e2, st));
        crag::FinParams fin;
        fin.corpus = ix->corpus;
        fin.inv_norm = ix->inv_norm;
        fin.a32 = (const float *)ws->a32.p;
        fin.qinv = (const float *)ws->qinv.p;
        fin.cand = (const uint2 *)ws->pf_cand.p;
        fin.count = (uint32_t *)ws->pf_count.p;
        fin.gbound = fp.gbound;
        fin.flags = (const uint32_t *)ws->pf_flags.p;
        fin.seq = fp.seq;
        fin.ids = ix->ids;
        fin.out_ids = d_out_ids;
        fin.out_scores = d_out_scores;
        fin.out_counts = d_out_counts;
        // statistics records: one block of PF_STAT_SLOTS / MAX_WS records per workspace, so that searches overlapping
        // on several streams never share a record (queries beyond a block's size fold onto it: counts may be lost
        // there, results never depend on them)
        fin.stats = ix->pf_stats + (size_t)(ws - ix->ws) * (crag::PF_STAT_SLOTS / crag_index::MAX_WS) * 3;
        fin.k = k;
        fin.cap = cap;
        fin.merge = mp;
        fin.nq = nq;
        fin.rsplit = rsplit;
        fin.xkeys = nullptr;
        fin.xids = nullptr;
        fin.xcount = nullptr;
        fin.xticket = nullptr;
        if (fin.rsplit > 1) {
            fin.xkeys = (uint64_t *)ws->pf_xkeys.p;
            fin.xids = (int64_t *)ws->pf_xids.p;
            fin.xcount = (uint2 *)ws->pf_xcount.p;
            fin.xticket = (uint32_t *)ws->pf_xticket.p;
        }
        // the fallback of a search whose candidate list overflows: the self-contained generic scan (32 queries per
        // pass) inside the same launch, see finalize_fb_kernel
        sp.wide = 0;
        sp.gate = nullptr;
        sp.unpipelined = 1;
        fin.scan = sp;
        fin.fb_blocks = G * ((nq + 31) / 32);
        fin.fb_done = (uint32_t *)ws->pf_flags.p + 1;
        fin.trace = ix->phase_trace;
        HIP_TRY(crag::launch_finalize(fin, st));
        ws->dirty = false;
    } else {
        HIP_TRY(crag::launch_scan(sp, q_blocks, st, &ix->last_scan_kernel));
        if (ev) HIP_TRY(hipEventRecord(ev->e2, st));
        HIP_TRY(crag::launch_merge_partials(mp, nq, st));
    }
    if (ev) HIP_TRY(hipEventRecord(ev->e3, st));
    // what a stream that later takes this workspace over waits for.  Recorded only once every workspace has an owner:
    // callers with up to MAX_WS streams (the pipelined form's three included) pay no event packet per search.
    ws->done_recorded = ix->record_done;
    if (ix->record_done) HIP_TRY(hipEventRecord(ws->done, st));
    return CRAG_OK;
}

int check_search_args(const crag_index *ix, const void *queries, int nq, int k, const void *mask,
                      int64_t mask_stride, const void *out_ids, const void *out_scores,
                      const void *out_counts) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    if (nq < 0) return fail(CRAG_EINVAL, "nq must be >= 0 (got %d)", nq);
    if (k <= 0 || k > CRAG_MAX_K) return fail(CRAG_EINVAL, "k must be in [1, %d] (got %d)", CRAG_MAX_K, k);
    if (nq > 0 && (!queries || !out_ids || !out_scores || !out_counts))
        return fail(CRAG_EINVAL, "queries / out_ids / out_scores / out_counts must not be NULL");
    if (mask) {
        const int64_t need = ((ix->size + 31) / 32) * 4;
        if (mask_stride != 0 && (mask_stride % 4 != 0 || mask_stride < need))
            return fail(CRAG_EINVAL, "mask_stride must be 0 or a multiple of 4 >= %lld (got %lld)",
                        (long long)need, (long long)mask_stride);
        if (((uintptr_t)mask) & 3) return fail(CRAG_EINVAL, "row_mask must be 4-byte aligned");
    }
    return CRAG_OK;
}

}  // namespace

extern "C" {

const char *crag_last_error(void) { return g_err; }

// used by the other translation units of the library (crag_encoder.hip) to report errors
void crag_set_error_(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

const char *crag_version(void) { return "cadence-rag_amd dense lane 0.1 (gfx950)"; }

int crag_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int crag_index_create(int device, int dim, int64_t capacity, crag_index **out) {
    if (!out) return fail(CRAG_EINVAL, "out is NULL");
    *out = nullptr;
    if (dim <= 0 || dim > CRAG_DIM) return fail(CRAG_EINVAL, "dim must be in [1, %d] (got %d)", CRAG_DIM, dim);
    if (capacity <= 0) return fail(CRAG_EINVAL, "capacity must be > 0 (got %lld)", (long long)capacity);
    int ndev = crag_device_count();
    if (ndev <= 0) return fail(CRAG_ENODEV, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(CRAG_EINVAL, "device %d out of range [0, %d)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(CRAG_EHIP, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CRAG_ENODEV, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    crag_index *ix = new (std::nothrow) crag_index();
    if (!ix) return fail(CRAG_ENOMEM, "out of host memory");
    ix->device = device;
    ix->dim = dim;
    ix->capacity = capacity;
    ix->cap_rows = ((capacity + 31) / 32) * 32;
    ix->n_cu = prop.multiProcessorCount;
    const size_t cbytes = (size_t)ix->cap_rows * crag::DIM * sizeof(float);
    hipError_t e;
    if ((e = hipMalloc((void **)&ix->corpus, cbytes)) != hipSuccess ||
        (e = hipMalloc((void **)&ix->inv_norm, (size_t)ix->cap_rows * sizeof(float))) != hipSuccess ||
        (e = hipMalloc((void **)&ix->ids, (size_t)ix->cap_rows * sizeof(int64_t))) != hipSuccess) {
        int rc = fail(CRAG_ENOMEM, "hipMalloc for %lld rows failed: %s", (long long)ix->cap_rows,
                      hipGetErrorString(e));
        crag_index_destroy(ix);
        return rc;
    }
    // padding rows of the last tile must read as "never eligible" and finite
    if ((e = hipMemset(ix->inv_norm, 0, (size_t)ix->cap_rows * sizeof(float))) != hipSuccess ||
        (e = hipMemset(ix->corpus, 0, cbytes)) != hipSuccess) {
        int rc = fail(CRAG_EHIP, "hipMemset failed: %s", hipGetErrorString(e));
        crag_index_destroy(ix);
        return rc;
    }
    for (auto &w : ix->ws)
        if ((e = hipEventCreateWithFlags(&w.done, hipEventDisableTiming)) != hipSuccess) {
            int rc = fail(CRAG_EHIP, "hipEventCreate failed: %s", hipGetErrorString(e));
            crag_index_destroy(ix);
            return rc;
        }
    ix->env_no_wide = getenv("CRAG_NO_WIDE") != nullptr;
    ix->env_no_reverse = getenv("CRAG_NO_REVERSE") != nullptr;
    ix->env_unpipelined = getenv("CRAG_UNPIPELINED") != nullptr;
    ix->env_no_prefilter = getenv("CRAG_NO_PREFILTER") != nullptr;
    ix->env_no_rsplit = getenv("CRAG_NO_RSPLIT") != nullptr;  // developer switch: one selection block per query for any k
    if (const char *v = getenv("CRAG_PF_NT")) ix->env_pf_nt = atoi(v) ? 1 : 0;
    if (const char *v = getenv("CRAG_PF_LAGS")) {
        int d = 0, r = 0;
        if (sscanf(v, "%d,%d", &d, &r) == 2 && d >= 1 && r > d && r <= 4) {
            ix->env_pf_derive_lag = d;
            ix->env_pf_read_lag = r;
        }
    }
    if (const char *v = getenv("CRAG_TEST_FAIL_AFTER_SCAN")) ix->env_fail_after_scan = atoll(v);
    if (const char *v = getenv("CRAG_PIPE_STREAMS")) {
        const int n = atoi(v);
        if (n >= 1 && n <= crag_index::MAX_PIPE) ix->n_pipe = n;
    }
    if (const char *v = getenv("CRAG_PF_NT_ABOVE_MB")) ix->nt_above_bytes = (int64_t)atoll(v) << 20;
    if (!ix->env_no_prefilter && getenv("CRAG_NO_FP16_MIRROR") == nullptr) {
        // + 2 KiB per row beside the 4 KiB fp32 row: the prefilter scan then streams half the bytes.  Padding rows
        // read as zeros (their positions are beyond every workgroup's row range anyway).
        const size_t mbytes = (size_t)ix->cap_rows * crag::DIM * sizeof(_Float16);
        if ((e = hipMalloc((void **)&ix->corpus16, mbytes)) != hipSuccess ||
            (e = hipMemset(ix->corpus16, 0, mbytes)) != hipSuccess) {
            int rc = fail(CRAG_ENOMEM, "hipMalloc for the fp16 mirror of %lld rows failed: %s", (long long)ix->cap_rows,
                          hipGetErrorString(e));
            crag_index_destroy(ix);
            return rc;
        }
    }
    if ((e = hipMalloc((void **)&ix->irregular_dev, sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&ix->pf_stats, crag::PF_STAT_SLOTS * 3 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(ix->irregular_dev, 0, sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(ix->pf_stats, 0, crag::PF_STAT_SLOTS * 3 * sizeof(unsigned long long))) != hipSuccess) {
        int rc = fail(CRAG_ENOMEM, "hipMalloc for the index state failed: %s", hipGetErrorString(e));
        crag_index_destroy(ix);
        return rc;
    }
    if (getenv("CRAG_PHASE_TRACE") != nullptr) {
        if ((e = hipMalloc((void **)&ix->phase_trace, 128 * sizeof(unsigned long long))) != hipSuccess ||
            (e = hipMemset(ix->phase_trace, 0, 128 * sizeof(unsigned long long))) != hipSuccess) {
            int rc = fail(CRAG_ENOMEM, "hipMalloc for the phase trace failed: %s", hipGetErrorString(e));
            crag_index_destroy(ix);
            return rc;
        }
    }
    *out = ix;
    return CRAG_OK;
}

int crag_index_destroy(crag_index *ix) {
    if (!ix) return CRAG_OK;
    DeviceGuard guard(ix->device);
    (void)hipDeviceSynchronize();
    for (auto &t : ix->ev_pool) {
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1a);
        (void)hipEventDestroy(t.e1);
        (void)hipEventDestroy(t.e2);
        (void)hipEventDestroy(t.e3);
    }
    if (ix->irregular_dev) (void)hipFree(ix->irregular_dev);
    if (ix->phase_trace) (void)hipFree(ix->phase_trace);
    for (int i = 0; i < crag_index::MAX_PIPE; ++i) {
        if (ix->pipe_fork[i]) (void)hipEventDestroy(ix->pipe_fork[i]);
        if (ix->pipe_done[i]) (void)hipEventDestroy(ix->pipe_done[i]);
        if (ix->pipe[i]) (void)hipStreamDestroy(ix->pipe[i]);
    }
    if (ix->pf_stats) (void)hipFree(ix->pf_stats);
    if (ix->corpus) (void)hipFree(ix->corpus);
    if (ix->corpus16) (void)hipFree(ix->corpus16);
    if (ix->inv_norm) (void)hipFree(ix->inv_norm);
    if (ix->ids) (void)hipFree(ix->ids);
    for (auto &w : ix->ws) {
        w.partial.release();
        w.gbound.release();
        w.a32.release();
        w.a16.release();
        w.qinv.release();
        w.pf_gbound.release();
        w.pf_cand.release();
        w.pf_count.release();
        w.pf_flags.release();
        w.pf_xkeys.release();
        w.pf_xids.release();
        w.pf_xcount.release();
        w.pf_xticket.release();
        if (w.done) (void)hipEventDestroy(w.done);
    }
    ix->stage_q.release();
    ix->stage_rows.release();
    ix->stage_ids.release();
    ix->stage_mask.release();
    ix->stage_out.release();
    ix->scratch.release();
    delete ix;
    return CRAG_OK;
}

int64_t crag_index_size(const crag_index *ix) { return ix ? ix->size : -1; }
int64_t crag_index_capacity(const crag_index *ix) { return ix ? ix->capacity : -1; }
int crag_index_dim(const crag_index *ix) { return ix ? ix->dim : -1; }

static int store_rows_locked(crag_index *ix, int64_t pos, const float *rows, int64_t n) {
    // chunked so that host staging stays bounded (64 Ki rows = 256 MiB at dim 1024)
    const int64_t CH = 65536;
    const bool dev = is_device_ptr(rows);
    for (int64_t o = 0; o < n; o += CH) {
        const int64_t m = (n - o < CH) ? (n - o) : CH;
        const float *src = rows + (size_t)o * ix->dim;
        if (!dev) {
            int rc = ix->stage_rows.ensure((size_t)m * ix->dim * sizeof(float));
            if (rc) return rc;
            HIP_TRY(hipMemcpy(ix->stage_rows.p, src, (size_t)m * ix->dim * sizeof(float), hipMemcpyHostToDevice));
            src = (const float *)ix->stage_rows.p;
        }
        HIP_TRY(crag::launch_store_rows(src, ix->dim, pos + o, m, ix->corpus, ix->inv_norm, ix->irregular_dev, ix->corpus16, 0));
        if (!dev) HIP_TRY(hipStreamSynchronize(0));  // staging buffer is reused by the next chunk
    }
    HIP_TRY(hipStreamSynchronize(0));
    if (!ix->irregular) {  // sticky: rows are never removed
        uint32_t flag = 0;
        HIP_TRY(hipMemcpy(&flag, ix->irregular_dev, sizeof(flag), hipMemcpyDeviceToHost));
        ix->irregular = flag != 0;
    }
    return CRAG_OK;
}

int crag_index_add(crag_index *ix, const float *rows, const int64_t *ids, int64_t n) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    if (n < 0) return fail(CRAG_EINVAL, "n must be >= 0");
    if (n == 0) return CRAG_OK;
    if (!rows) return fail(CRAG_EINVAL, "rows is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->size + n > ix->capacity)
        return fail(CRAG_ENOMEM, "capacity exceeded: size %lld + %lld > %lld", (long long)ix->size,
                    (long long)n, (long long)ix->capacity);
    if (ix->size + n >= (int64_t)0xfffffff0ll) return fail(CRAG_ENOMEM, "more than 2^32 rows per index");
    DeviceGuard guard(ix->device);
    const int64_t pos = ix->size;
    // ids must grow with the row position: that is what makes "equal scores by ascending position" inside
    // the scan the same order as "equal scores by ascending id"  and as the cross-shard merge
    int64_t new_last;
    if (!ids) {
        if (pos <= ix->last_id)
            return fail(CRAG_EINVAL, "implicit ids would start at %lld, not above the largest stored id %lld",
                        (long long)pos, (long long)ix->last_id);
        new_last = pos + n - 1;
    } else if (is_device_ptr(ids)) {
        int rc0 = ix->scratch.ensure(sizeof(unsigned long long));
        if (rc0) return rc0;
        HIP_TRY(hipMemsetAsync(ix->scratch.p, 0, sizeof(unsigned long long), 0));
        HIP_TRY(crag::launch_check_ids(ids, n, ix->last_id, (unsigned long long *)ix->scratch.p, 0));
        unsigned long long bad = 0;
        HIP_TRY(hipMemcpy(&bad, ix->scratch.p, sizeof(bad), hipMemcpyDeviceToHost));
        if (bad)
            return fail(CRAG_EINVAL, "ids must be strictly ascending and above the largest stored id %lld "
                        "(%llu of %lld are not)", (long long)ix->last_id, bad, (long long)n);
        HIP_TRY(hipMemcpy(&new_last, ids + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost));
    } else {
        int64_t prev = ix->last_id;
        for (int64_t i = 0; i < n; ++i) {
            if (ids[i] <= prev)
                return fail(CRAG_EINVAL, "ids must be strictly ascending and above the largest stored id: "
                            "ids[%lld] = %lld follows %lld", (long long)i, (long long)ids[i], (long long)prev);
            prev = ids[i];
        }
        new_last = prev;
    }
    int rc = store_rows_locked(ix, pos, rows, n);
    if (rc) return rc;
    if (ids) {
        HIP_TRY(hipMemcpy(ix->ids + pos, ids, (size_t)n * sizeof(int64_t),
                          is_device_ptr(ids) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    } else {
        HIP_TRY(crag::launch_fill_ids(ix->ids, pos, n, pos, 0));
        HIP_TRY(hipStreamSynchronize(0));
    }
    ix->size = pos + n;
    ix->last_id = new_last;
    return CRAG_OK;
}

int crag_index_update(crag_index *ix, int64_t pos, const float *rows, int64_t n) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    if (n < 0 || pos < 0) return fail(CRAG_EINVAL, "pos and n must be >= 0");
    if (n == 0) return CRAG_OK;
    if (!rows) return fail(CRAG_EINVAL, "rows is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (pos + n > ix->size)
        return fail(CRAG_EINVAL, "update range [%lld, %lld) exceeds size %lld", (long long)pos,
                    (long long)(pos + n), (long long)ix->size);
    DeviceGuard guard(ix->device);
    return store_rows_locked(ix, pos, rows, n);
}

int crag_index_get_rows(crag_index *ix, int64_t pos, int64_t n, float *rows, int64_t *ids) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    if (n < 0 || pos < 0) return fail(CRAG_EINVAL, "pos and n must be >= 0");
    if (n == 0) return CRAG_OK;
    if (!rows) return fail(CRAG_EINVAL, "rows is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (pos + n > ix->size)
        return fail(CRAG_EINVAL, "range [%lld, %lld) exceeds size %lld", (long long)pos,
                    (long long)(pos + n), (long long)ix->size);
    DeviceGuard guard(ix->device);
    const bool dev = is_device_ptr(rows);
    const int64_t CH = 65536;
    for (int64_t o = 0; o < n; o += CH) {
        const int64_t m = (n - o < CH) ? (n - o) : CH;
        float *dst = rows + (size_t)o * ix->dim;
        float *ddst = dst;
        if (!dev) {
            int rc = ix->stage_rows.ensure((size_t)m * ix->dim * sizeof(float));
            if (rc) return rc;
            ddst = (float *)ix->stage_rows.p;
        }
        HIP_TRY(crag::launch_load_rows(ix->corpus, ix->dim, pos + o, m, ddst, crag::crag_piece_shift(ix->corpus16 != nullptr), 0));
        if (!dev)
            HIP_TRY(hipMemcpy(dst, ddst, (size_t)m * ix->dim * sizeof(float), hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipStreamSynchronize(0));
    if (ids)
        HIP_TRY(hipMemcpy(ids, ix->ids + pos, (size_t)n * sizeof(int64_t),
                          is_device_ptr(ids) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return CRAG_OK;
}

int crag_index_count_eligible(crag_index *ix, const uint8_t *row_mask, int64_t *out_count) {
    if (!ix || !out_count) return fail(CRAG_EINVAL, "index / out_count is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);
    *out_count = 0;
    if (ix->size == 0) return CRAG_OK;
    const uint32_t *dmask = nullptr;
    if (row_mask) {
        if (((uintptr_t)row_mask) & 3) return fail(CRAG_EINVAL, "row_mask must be 4-byte aligned");
        const size_t mbytes = (size_t)((ix->size + 31) / 32) * 4;
        if (is_device_ptr(row_mask)) {
            dmask = (const uint32_t *)row_mask;
        } else {
            int rc = ix->stage_mask.ensure(mbytes);
            if (rc) return rc;
            // the caller's buffer may end at ceil(size/8) bytes: copy only that much
            const size_t have = (size_t)((ix->size + 7) / 8);
            HIP_TRY(hipMemset(ix->stage_mask.p, 0, mbytes));
            HIP_TRY(hipMemcpy(ix->stage_mask.p, row_mask, have, hipMemcpyHostToDevice));
            dmask = (const uint32_t *)ix->stage_mask.p;
        }
    }
    int rc = ix->scratch.ensure(sizeof(unsigned long long));
    if (rc) return rc;
    HIP_TRY(hipMemset(ix->scratch.p, 0, sizeof(unsigned long long)));
    HIP_TRY(crag::launch_count_eligible(ix->inv_norm, ix->size, dmask, (unsigned long long *)ix->scratch.p, 0));
    unsigned long long c = 0;
    HIP_TRY(hipMemcpy(&c, ix->scratch.p, sizeof(c), hipMemcpyDeviceToHost));
    *out_count = (int64_t)c;
    return CRAG_OK;
}

int crag_index_search_async(crag_index *ix, const float *d_queries, int nq, int k,
                            const uint8_t *d_row_mask, int64_t mask_stride, int64_t *d_out_ids,
                            float *d_out_scores, int32_t *d_out_counts, void *stream) {
    int rc = check_search_args(ix, d_queries, nq, k, d_row_mask, mask_stride, d_out_ids, d_out_scores,
                               d_out_counts);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);
    return search_device(ix, d_queries, nq, k, d_row_mask, mask_stride, d_out_ids, d_out_scores,
                         d_out_counts, (hipStream_t)stream);
}

int crag_index_search_pipelined(crag_index *ix, const float *d_queries, int nq, int k,
                                const uint8_t *d_row_mask, int64_t mask_stride, int64_t *d_out_ids,
                                float *d_out_scores, int32_t *d_out_counts, void *stream, int flags) {
    int rc = check_search_args(ix, d_queries, nq, k, d_row_mask, mask_stride, d_out_ids, d_out_scores,
                               d_out_counts);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);
    for (int i = 0; i < ix->n_pipe; ++i) {   // created on first use; a creation that failed is tried again, not skipped
        if (!ix->pipe[i]) HIP_TRY(hipStreamCreateWithFlags(&ix->pipe[i], hipStreamNonBlocking));
        if (!ix->pipe_fork[i]) HIP_TRY(hipEventCreateWithFlags(&ix->pipe_fork[i], hipEventDisableTiming));
        if (!ix->pipe_done[i]) HIP_TRY(hipEventCreateWithFlags(&ix->pipe_done[i], hipEventDisableTiming));
    }
    // Overlap pays for the searches whose small kernels are a large share of the step -- k <= 24 (one class set): 100 000
    // x 64, k = 10: 40.2 us per step on three streams against 47.7 in order; 1M: 316 against 322 -- and costs for larger k,
    // whose scans disturb each other's bound exchange (k = 100 at 100 000 rows: 74-80 us against 71).  Those run in
    // stream order on the caller's stream (the join then has nothing to wait for).
    if (k > 24 || ix->n_pipe <= 1)
        return search_device(ix, d_queries, nq, k, d_row_mask, mask_stride, d_out_ids, d_out_scores, d_out_counts,
                             (hipStream_t)stream);
    const int i = (int)(ix->pipe_next++ % (unsigned)ix->n_pipe);
    if (!(flags & CRAG_PIPE_INPUTS_READY)) {
        HIP_TRY(hipEventRecord(ix->pipe_fork[i], (hipStream_t)stream));
        HIP_TRY(hipStreamWaitEvent(ix->pipe[i], ix->pipe_fork[i], 0));
    }
    rc = search_device(ix, d_queries, nq, k, d_row_mask, mask_stride, d_out_ids, d_out_scores, d_out_counts,
                       ix->pipe[i]);
    if (rc) return rc;
    ix->pipe_pending[i] = true;   // (its completion event is recorded by the join: one per fence and stream, not per search)
    return CRAG_OK;
}

int crag_index_join(crag_index *ix, void *stream) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);
    for (int i = 0; i < ix->n_pipe; ++i)
        if (ix->pipe_pending[i]) {
            HIP_TRY(hipEventRecord(ix->pipe_done[i], ix->pipe[i]));
            HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ix->pipe_done[i], 0));
            ix->pipe_pending[i] = false;
        }
    return CRAG_OK;
}

int crag_index_search(crag_index *ix, const float *queries, int nq, int k, const uint8_t *row_mask,
                      int64_t mask_stride, int64_t *out_ids, float *out_scores, int32_t *out_counts) {
    int rc = check_search_args(ix, queries, nq, k, row_mask, mask_stride, out_ids, out_scores, out_counts);
    if (rc) return rc;
    if (nq == 0) return CRAG_OK;
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);

    const float *dq = queries;
    if (!is_device_ptr(queries)) {
        const size_t b = (size_t)nq * ix->dim * sizeof(float);
        if ((rc = ix->stage_q.ensure(b))) return rc;
        HIP_TRY(hipMemcpy(ix->stage_q.p, queries, b, hipMemcpyHostToDevice));
        dq = (const float *)ix->stage_q.p;
    }
    const uint8_t *dm = row_mask;
    int64_t dstride = mask_stride;
    if (row_mask && !is_device_ptr(row_mask)) {
        const size_t words = (size_t)((ix->size + 31) / 32);
        const size_t row_b = words * 4;
        const int nmask = mask_stride ? nq : 1;
        if ((rc = ix->stage_mask.ensure(row_b * nmask))) return rc;
        HIP_TRY(hipMemset(ix->stage_mask.p, 0, row_b * nmask));
        const size_t have = (size_t)((ix->size + 7) / 8);
        if (mask_stride == 0) {
            HIP_TRY(hipMemcpy(ix->stage_mask.p, row_mask, have, hipMemcpyHostToDevice));
        } else {
            HIP_TRY(hipMemcpy2D(ix->stage_mask.p, row_b, row_mask, (size_t)mask_stride, have, nq,
                                hipMemcpyHostToDevice));
            dstride = (int64_t)row_b;
        }
        dm = (const uint8_t *)ix->stage_mask.p;
    }
    const bool ids_dev = is_device_ptr(out_ids), sc_dev = is_device_ptr(out_scores),
               ct_dev = is_device_ptr(out_counts);
    const size_t b_ids = (size_t)nq * k * sizeof(int64_t), b_sc = (size_t)nq * k * sizeof(float),
                 b_ct = (size_t)nq * sizeof(int32_t);
    if ((rc = ix->stage_out.ensure(b_ids + b_sc + b_ct + 64))) return rc;
    char *so = (char *)ix->stage_out.p;
    int64_t *d_ids = ids_dev ? out_ids : (int64_t *)so;
    float *d_sc = sc_dev ? out_scores : (float *)(so + b_ids);
    int32_t *d_ct = ct_dev ? out_counts : (int32_t *)(so + b_ids + b_sc);

    rc = search_device(ix, dq, nq, k, dm, dstride, d_ids, d_sc, d_ct, 0);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(0));
    if (!ids_dev) HIP_TRY(hipMemcpy(out_ids, d_ids, b_ids, hipMemcpyDeviceToHost));
    if (!sc_dev) HIP_TRY(hipMemcpy(out_scores, d_sc, b_sc, hipMemcpyDeviceToHost));
    if (!ct_dev) HIP_TRY(hipMemcpy(out_counts, d_ct, b_ct, hipMemcpyDeviceToHost));
    return CRAG_OK;
}

int crag_merge_topk(int device, const int64_t *d_ids, const float *d_scores, const int32_t *d_counts,
                    int n_lists, int nq, int k, int64_t *d_out_ids, float *d_out_scores,
                    int32_t *d_out_counts, void *stream) {
    if (!d_ids || !d_scores || !d_counts || !d_out_ids || !d_out_scores || !d_out_counts)
        return fail(CRAG_EINVAL, "NULL pointer argument");
    if (n_lists <= 0 || nq < 0 || k <= 0 || k > CRAG_MAX_K)
        return fail(CRAG_EINVAL, "bad sizes n_lists=%d nq=%d k=%d", n_lists, nq, k);
    if ((int64_t)n_lists * k > 4096) return fail(CRAG_EINVAL, "n_lists*k must be <= 4096");
    if (nq == 0) return CRAG_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(CRAG_EHIP, "hipSetDevice(%d) failed", device);
    crag::XMergeParams p;
    p.ids = d_ids;
    p.scores = d_scores;
    p.counts = d_counts;
    p.out_ids = d_out_ids;
    p.out_scores = d_out_scores;
    p.out_counts = d_out_counts;
    p.stride_ids = (int64_t)nq * k;
    p.stride_scores = (int64_t)nq * k;
    p.stride_counts = nq;
    p.n_lists = n_lists;
    p.nq = nq;
    p.k = k;
    HIP_TRY(crag::launch_merge_results(p, (hipStream_t)stream));
    return CRAG_OK;
}

int64_t crag_result_record_bytes(int nq, int k) {
    if (nq < 0 || k <= 0) return -1;
    const int64_t b = (int64_t)nq * k * 8 + (int64_t)nq * k * 4 + (int64_t)nq * 4;
    return (b + 7) & ~(int64_t)7;
}

int crag_merge_topk_packed(int device, const void *d_records, int n_lists, int nq, int k, int64_t *d_out_ids,
                           float *d_out_scores, int32_t *d_out_counts, void *stream) {
    if (!d_records || !d_out_ids || !d_out_scores || !d_out_counts) return fail(CRAG_EINVAL, "NULL pointer argument");
    if (n_lists <= 0 || nq < 0 || k <= 0 || k > CRAG_MAX_K)
        return fail(CRAG_EINVAL, "bad sizes n_lists=%d nq=%d k=%d", n_lists, nq, k);
    if ((int64_t)n_lists * k > 4096) return fail(CRAG_EINVAL, "n_lists*k must be <= 4096");
    if (nq == 0) return CRAG_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(CRAG_EHIP, "hipSetDevice(%d) failed", device);
    const int64_t rec = crag_result_record_bytes(nq, k);
    const char *base = (const char *)d_records;
    crag::XMergeParams p;
    p.ids = (const int64_t *)base;
    p.scores = (const float *)(base + (int64_t)nq * k * 8);
    p.counts = (const int32_t *)(base + (int64_t)nq * k * 12);
    p.out_ids = d_out_ids;
    p.out_scores = d_out_scores;
    p.out_counts = d_out_counts;
    p.stride_ids = rec / 8;
    p.stride_scores = rec / 4;
    p.stride_counts = rec / 4;
    p.n_lists = n_lists;
    p.nq = nq;
    p.k = k;
    HIP_TRY(crag::launch_merge_results(p, (hipStream_t)stream));
    return CRAG_OK;
}

int crag_index_profile_enable(crag_index *ix, int enabled) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->profiling = enabled > 0 ? enabled : 0;
    ix->prof_calls = 0;
    ix->ev_used = 0;
    return CRAG_OK;
}

int crag_index_profile_read_ex(crag_index *ix, int64_t *n_launches, double *scan_ms_total,
                               double *merge_ms_total, double *event_pair_ms_total) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);
    double scan = 0.0, merge = 0.0, pair = 0.0;
    for (size_t i = 0; i < ix->ev_used; ++i) {
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
        HIP_TRY(hipEventSynchronize(ix->ev_pool[i].e3));
        HIP_TRY(hipEventElapsedTime(&a, ix->ev_pool[i].e0, ix->ev_pool[i].e1a));
        HIP_TRY(hipEventElapsedTime(&d, ix->ev_pool[i].e1a, ix->ev_pool[i].e1));
        HIP_TRY(hipEventElapsedTime(&b, ix->ev_pool[i].e1, ix->ev_pool[i].e2));
        HIP_TRY(hipEventElapsedTime(&c, ix->ev_pool[i].e2, ix->ev_pool[i].e3));
        scan += b;
        merge += a + c;
        pair += d;
    }
    if (n_launches) *n_launches = (int64_t)ix->ev_used;
    if (scan_ms_total) *scan_ms_total = scan;
    if (merge_ms_total) *merge_ms_total = merge;
    if (event_pair_ms_total) *event_pair_ms_total = pair;
    ix->ev_used = 0;
    return CRAG_OK;
}

int crag_index_profile_read(crag_index *ix, int64_t *n_launches, double *scan_ms_total,
                            double *merge_ms_total) {
    return crag_index_profile_read_ex(ix, n_launches, scan_ms_total, merge_ms_total, nullptr);
}

int crag_index_prefilter_stats(crag_index *ix, int64_t *searches, int64_t *candidates, int64_t *rescored_rows) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);
    unsigned long long v[3] = {0, 0, 0};
    std::vector<unsigned long long> rec((size_t)crag::PF_STAT_SLOTS * 3);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rec.data(), ix->pf_stats, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ix->pf_stats, 0, rec.size() * sizeof(unsigned long long)));
    for (size_t i = 0; i < rec.size(); ++i) v[i % 3] += rec[i];
    if (candidates) *candidates = (int64_t)v[0];
    if (rescored_rows) *rescored_rows = (int64_t)v[1];
    if (searches) *searches = (int64_t)v[2];
    return CRAG_OK;
}

int crag_index_phase_trace(crag_index *ix, uint64_t *out128) {
    if (!ix || !out128) return fail(CRAG_EINVAL, "index / out is NULL");
    if (!ix->phase_trace) return fail(CRAG_EINVAL, "the index was not created with CRAG_PHASE_TRACE=1");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard guard(ix->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out128, ix->phase_trace, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CRAG_OK;
}

const char *crag_index_last_scan_kernel(const crag_index *ix) { return ix ? ix->last_scan_kernel : ""; }

int64_t crag_index_prefilter_row_bytes(const crag_index *ix) {
    if (!ix || ix->env_no_prefilter) return 0;
    return (int64_t)crag::DIM * (ix->corpus16 ? 2 : 4);
}

int crag_index_scan_geometry(const crag_index *ix, int nq, int *workgroups, int *threads,
                             int *query_blocks, int64_t *algorithmic_bytes_per_launch) {
    if (!ix) return fail(CRAG_EINVAL, "index is NULL");
    const int qb = (nq + 31) / 32;
    if (workgroups) *workgroups = scan_groups(ix);
    if (threads) *threads = crag::SCAN_THREADS;
    if (query_blocks) *query_blocks = qb;
    // N*D*4 (corpus streamed once per query batch) + Q*D*4.  A batch of more than
    // 32 queries re-streams the corpus once per 32-query block; that is NOT counted here.
    if (algorithmic_bytes_per_launch)
        *algorithmic_bytes_per_launch = ix->size * (int64_t)ix->dim * 4 + (int64_t)nq * ix->dim * 4;
    return CRAG_OK;
}

}  // extern "C"
