/*
 * crag_dense.h — C ABI of the MI355X-native dense-retrieval lane (libcrag_dense.so).
 *
 * This is the drop-in boundary for the reference's dense path.  The reference has no FFI
 * today: its "interface" for this path is (a) the pgvector SQL issued by
 *   _fetch_chunks_dense        /root/reference/app/retrieve.py:326-354
 *   _fetch_artifacts_dense     /root/reference/app/retrieve.py:357-389
 *   _estimate_dense_candidates /root/reference/app/retrieve.py:303-323
 * and the per-row `UPDATE … SET embedding = CAST(:lit AS vector(1024))` of
 *   _update_embeddings         /root/reference/app/embedding_pipeline.py:149-168
 * and (b) the HTTP embedding gateway behind embed_texts
 *                              /root/reference/app/embeddings.py:48-82
 * (gateway math: P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:683-716).
 * Each entry point below names the reference interface it replaces.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success and a negative CRAG_E* code on error
 * (never throws); crag_last_error() returns a thread-local message for the last failure on
 * the calling thread.  Plain pointers and sizes only — no torch / HIP types.  "dev" pointers
 * are device (HBM) addresses on the index's device; `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  The library owns the device corpus.
 */
#ifndef CRAG_DENSE_H
#define CRAG_DENSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRAG_OK 0
#define CRAG_EINVAL (-1)   /* bad argument */
#define CRAG_EHIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define CRAG_ENOMEM (-3)   /* capacity exceeded / allocation failed */
#define CRAG_E2BIG (-4)    /* an input exceeds what one call takes; the caller splits it (crag_tech_lane_host) */
#define CRAG_ENODEV (-4)   /* no usable gfx950 device */

#define CRAG_MAX_K 128     /* reference uses k = 50 / 10 (retrieve.py:18-19); BASELINE asks 10..100 */
#define CRAG_DIM 1024      /* vector(1024): alembic/versions/0001_initial_schema.py:87 */

typedef struct crag_index crag_index;

const char *crag_last_error(void);
const char *crag_version(void);
/* Number of HIP devices visible (0 if none / no driver). */
int crag_device_count(void);

/* ---- corpus: replaces the pgvector `embedding vector(1024)` column + its scan ---------- */

/* Allocate an empty index for up to `capacity` rows of `dim` (<= 1024) floats on `device`.
 * Replaces: the table column + HNSW/seq-scan storage (0001_initial_schema.py:87,98-102).
 * HBM per row: 4 KiB (the fp32 row, source of truth) + 2 KiB (fp16 mirror of the unit row, what the prefilter scan
 * streams) + 12 bytes.  Environment, read once here: CRAG_NO_FP16_MIRROR=1 leaves the mirror out (the prefilter scan
 * then streams the fp32 rows: twice the bytes per search, two thirds of the footprint); CRAG_NO_PREFILTER=1 keeps
 * every search on the exact fp32 MFMA scan.  Results are bit-identical in all three modes.  (Developer switches,
 * same place: CRAG_PF_NT=0/1 and CRAG_PF_NT_ABOVE_MB=<n> override when the mirror scan uses the streaming cache
 * policy -- by default for mirrors above 1.5 GB.) */
int crag_index_create(int device, int dim, int64_t capacity, crag_index **out);
int crag_index_destroy(crag_index *ix);

/* Append n rows ([n, dim] row-major fp32; host OR device pointer, detected) with their
 * 64-bit ids (NULL => consecutive ids continuing from the current size).  ids must be strictly
 * ascending and above every id already stored (CRAG_EINVAL otherwise, nothing is stored): the
 * backfill feeds rows `ORDER BY id` (embedding_pipeline.py:136), and this is what makes the search
 * order below "descending score, then ascending id".  A row embedded late (its id below the stored
 * maximum) goes in through a rebuild: cadence_rag_amd.retrieve.DenseTable.insert does that.
 * Rows with a zero or non-finite norm are stored but never returned (pgvector gives them a NaN
 * distance).
 * Replaces: _update_embeddings' per-row UPDATE (embedding_pipeline.py:157-168).
 * Appending is safe while searches enqueued earlier with crag_index_search_async are still running (they
 * never read past the size they were launched with); crag_index_update rewrites rows in place and must
 * not overlap in time with searches in flight on other streams -- those of crag_index_search_pipelined run on
 * streams of the index's own: crag_index_join + a synchronisation of the joined stream come first. */
int crag_index_add(crag_index *ix, const float *rows, const int64_t *ids, int64_t n);

/* Overwrite the vectors of rows [pos, pos+n) (positions, not ids) — re-embed in place. */
int crag_index_update(crag_index *ix, int64_t pos, const float *rows, int64_t n);

/* Rows currently stored / capacity / dim. */
int64_t crag_index_size(const crag_index *ix);
int64_t crag_index_capacity(const crag_index *ix);
int crag_index_dim(const crag_index *ix);

/* Copy rows [pos, pos+n) back out as [n, dim] row-major fp32 (host or device pointer) and,
 * if ids != NULL, their ids.  Bit-exact with what was added (the corpus is stored raw). */
int crag_index_get_rows(crag_index *ix, int64_t pos, int64_t n, float *rows, int64_t *ids);

/* Number of rows that can be returned for a (shared, nullable) mask: rows with a finite
 * non-zero norm whose mask bit is set.  Replaces: _estimate_dense_candidates' COUNT(*)
 * (retrieve.py:303-323).  mask: host or device pointer. */
int crag_index_count_eligible(crag_index *ix, const uint8_t *row_mask, int64_t *out_count);

/* Exact cosine top-k, synchronous; every pointer may be host or device (detected).
 *   queries     [nq, dim] row-major fp32 (need not be normalised)
 *   row_mask    nullable.  Bit (i & 7) of byte row_mask[q*mask_stride + (i >> 3)] set =>
 *               row at position i is eligible for query q.  mask_stride = 0 => one mask
 *               shared by all queries; otherwise a multiple of 4 bytes >= ceil(size/32)*4.
 *               Base address 4-byte aligned.  (Encodes _build_filter_clause, retrieve.py:93-120.)
 *   out_ids     [nq, k]  best first; -1 padded
 *   out_scores  [nq, k]  cosine similarity = 1 - (embedding <=> q), clamped to [-1, 1]; NaN padded
 *   out_counts  [nq]     valid entries per query (<= k)
 * Order: descending score, equal scores by ascending id (crag_index_add keeps ids
 * ascending with the row position, so the scan breaks ties on the position).  The reference SQL has
 * no tie-break at all.
 * Replaces: _fetch_chunks_dense / _fetch_artifacts_dense `ORDER BY embedding <=> q LIMIT k`
 * (retrieve.py:339-353, 369-388). */
int crag_index_search(crag_index *ix, const float *queries, int nq, int k,
                      const uint8_t *row_mask, int64_t mask_stride, int64_t *out_ids,
                      float *out_scores, int32_t *out_counts);

/* Same, all pointers DEVICE, enqueued on `stream` with no host synchronisation (the form
 * bench.py, the multi-GPU lane and hipGraph capture use). */
int crag_index_search_async(crag_index *ix, const float *d_queries, int nq, int k,
                            const uint8_t *d_row_mask, int64_t mask_stride, int64_t *d_out_ids,
                            float *d_out_scores, int32_t *d_out_counts, void *stream);

/* Throughput form of crag_index_search_async for ONE caller stream that issues a run of INDEPENDENT searches (a
 * batch job: the eval harness, bulk re-ranking, bench.py): consecutive calls take turns on three streams the index
 * owns (CRAG_PIPE_STREAMS=1..4) -- each with its own workspace --, so that the small kernels of one search (query
 * preparation, selection) run beside the scan of another; an in-order stream leaves most of the chip idle during them
 * (12 of 48 us per search at 100 000 rows x 64 queries, k = 10: 40 us per step pipelined).  Searches with k > 24 gain
 * nothing from it (their scans disturb each other's bound exchange) and run in stream order on the caller's stream.  What the caller gives up is the stream order between a search and what follows:
 *   * the OUTPUTS of a call are defined on `stream` only behind crag_index_join(ix, stream) (which makes `stream`
 *     wait for every pipelined search issued so far; it does not block the host);
 *   * flags & CRAG_PIPE_INPUTS_READY: the caller states that queries / row_mask are complete in memory when the call
 *     is made (resident inputs); without it the library orders its internal stream behind everything enqueued on
 *     `stream` so far (one event per call).
 * Results are the same bits as crag_index_search_async's.  Same reference call sites as crag_index_search
 * (retrieve.py:339-353, 369-388); the reference itself runs one query per request and has no counterpart. */
#define CRAG_PIPE_INPUTS_READY 1
int crag_index_search_pipelined(crag_index *ix, const float *d_queries, int nq, int k,
                                const uint8_t *d_row_mask, int64_t mask_stride, int64_t *d_out_ids,
                                float *d_out_scores, int32_t *d_out_counts, void *stream, int flags);
int crag_index_join(crag_index *ix, void *stream);

/* Merge per-shard results (the multi-GPU exchange step: each rank's [nq, k] top-k after an
 * RCCL all-gather) into the global top-k.  All pointers DEVICE.
 *   d_ids/d_scores/d_counts  [n_lists, nq, k] / [n_lists, nq, k] / [n_lists, nq]
 * Same ordering rule as crag_index_search (score desc, id asc). */
int crag_merge_topk(int device, const int64_t *d_ids, const float *d_scores,
                    const int32_t *d_counts, int n_lists, int nq, int k, int64_t *d_out_ids,
                    float *d_out_scores, int32_t *d_out_counts, void *stream);

/* One-collective form of the exchange step.  A "result record" holds one rank's search output as
 * [ids nq*k int64][scores nq*k fp32][counts nq int32], padded to a multiple of 8 bytes
 * (crag_result_record_bytes).  Each rank lets crag_index_search_async write straight into its
 * record, ONE all-gather moves the records, and this merges n_lists consecutive records. */
int64_t crag_result_record_bytes(int nq, int k);
int crag_merge_topk_packed(int device, const void *d_records, int n_lists, int nq, int k,
                           int64_t *d_out_ids, float *d_out_scores, int32_t *d_out_counts, void *stream);

/* Reciprocal-rank fusion of up to 8 retrieval lanes on the GPU (hybrid /retrieve, BASELINE configs[4]).
 * Replaces: _rrf_merge (retrieve.py:245-260) — score += 1/(rrf_k + rank) per lane in lane order (fp64,
 * bit-identical to the Python floats), stable descending order (ties keep first-insertion order).
 *   d_lane_ids[l]    device [nq, lane_width[l]] int64 keys of lane l, best first
 *   d_lane_counts[l] device [nq] valid entries per query
 *   outputs          [nq, out_k]: fused keys (-1 pad), fp64 scores (NaN pad), lane-hit bit masks; [nq] counts
 * The pointer arrays themselves live on the HOST. */
int crag_rrf_fuse(int n_lanes, const int64_t *const *d_lane_ids, const int32_t *const *d_lane_counts,
                  const int *lane_width, int nq, int rrf_k, int out_k, int64_t *d_out_ids,
                  double *d_out_scores, uint32_t *d_out_lanes, int32_t *d_out_counts, void *stream);

/* Exact-token lane for a batch of up to 64 queries (hybrid /retrieve).  Replaces: _fetch_chunks_tech /
 * _fetch_artifacts_tech (retrieve.py:183-242): rows whose token set overlaps the query's, first k in
 * the order (call_started_at DESC, id ASC).  Tokens are 64-bit hashes of the exact token strings.
 *   d_order [n] int32   row position at each rank r of that static order
 *   d_row_ptr [n+1] int64, d_tokens [nnz] uint64   CSR of the rows' token hashes, stored BY RANK
 *                       (CSR row r = the row at position d_order[r]) so the scan streams coalesced
 *   d_query_tokens [nq, 32] uint64, d_query_token_counts [nq] int32 (<= 32 tokens per query)
 *   d_row_mask as in crag_index_search (bit per row POSITION; nullable)
 *   d_bitmap_scratch [ceil(n/64) * nq] uint64
 *   d_out_ids [nq, k] (-1 pad), d_out_counts [nq] */
int crag_tech_lane(const int32_t *d_order, const int64_t *d_row_ptr, const uint64_t *d_tokens,
                   const int64_t *d_ids, int64_t n_rows, const uint64_t *d_query_tokens,
                   const int32_t *d_query_token_counts, int nq, int k, const uint8_t *d_row_mask,
                   int64_t mask_stride, uint64_t *d_bitmap_scratch, int64_t *d_out_ids,
                   int32_t *d_out_counts, void *stream);

/* The same lane for a caller that holds the query tokens on the HOST (the gateway does: extract_tech_tokens runs there).
 * An upload slot = a pinned host buffer, its device twin and the event of the last copy that read the host buffer; a
 * caller keeps a small ring of them per stream (the call waits for the slot's previous copy only).  The call packs the
 * hashes (duplicates inside a query dropped, first occurrence kept), enqueues ONE host-to-device copy and the two
 * kernels on `stream` and returns.
 *   h_token_hashes  host: the queries' token hashes back to back;  h_token_counts host [nq]: tokens per query
 *   CRAG_E2BIG: a query holds more than 32 DISTINCT tokens -- nothing was enqueued, the caller runs it in passes. */
typedef struct crag_upload_slot crag_upload_slot;
crag_upload_slot *crag_upload_slot_create(void);
void crag_upload_slot_destroy(crag_upload_slot *slot);
int crag_tech_lane_host(const int32_t *d_order, const int64_t *d_row_ptr, const uint64_t *d_tokens,
                        const int64_t *d_ids, int64_t n_rows, const uint64_t *h_token_hashes,
                        const int32_t *h_token_counts, int nq, int k, const uint8_t *d_row_mask, int64_t mask_stride,
                        crag_upload_slot *slot, uint64_t *d_bitmap_scratch, int64_t *d_out_ids, int32_t *d_out_counts,
                        void *stream);

/* Live kernel timing for bench.py's roofline: enabled = N > 0 records HIP events around the scan
 * (and merge) kernel of every N-th search, on the stream it is launched on (N = 1: every search;
 * larger N perturbs the timed region less); 0 disables.  crag_index_profile_read sums and clears
 * the recorded samples (synchronises the events); n_launches = number of samples; scan_ms_total = the scan
 * kernel alone, merge_ms_total = everything else of a search (query preparation, rescoring / merge). */
int crag_index_profile_enable(crag_index *ix, int enabled);
int crag_index_profile_read(crag_index *ix, int64_t *n_launches, double *scan_ms_total,
                            double *merge_ms_total);
/* The same + event_pair_ms_total: every sample also records two events back to back in front of the scan launch;
 * their elapsed time is what an event pair measures with NOTHING between (the part of scan_ms_total that is event
 * processing, not kernel: rocprofv3's begin/end timestamps of the kernel do not contain it). */
int crag_index_profile_read_ex(crag_index *ix, int64_t *n_launches, double *scan_ms_total,
                               double *merge_ms_total, double *event_pair_ms_total);

/* Byte accounting of the prefilter path (fp16 MFMA scan + exact fp32 rescoring of the candidates, the path
 * searches over corpora of >= 128 rows per workgroup take): sums since the last call, then cleared.
 * candidates = rows that passed the proven-bound filter, rescored_rows = rows re-read (4 KiB each) for the
 * exact fp32 score.  The records are kept per workspace (= per stream in use, up to eight) and per query, written with
 * plain stores by the one selection block that owns them: searches overlapping on different streams do not lose counts;
 * a search of more than 256 queries folds its queries onto 256 records and may (results never depend on it).  A search
 * answered by the overflow fallback is not counted.  Synchronises the device. */
int crag_index_prefilter_stats(crag_index *ix, int64_t *searches, int64_t *candidates, int64_t *rescored_rows);

/* Developer probe (index created with CRAG_PHASE_TRACE=1 in the environment, else CRAG_EINVAL): 128 words written by
 * the selection blocks of query 0 of the most recent prefilter search -- [block r of the query][16]: 100 MHz
 * timestamps at the phase boundaries (0 start, 1 candidates loaded, 2 k-th approximate score, 3 survivors rescored,
 * 4 own list written + ticket, 5 lists gathered, 6 end), [8] candidates, [9] rows this block rescored.
 * Synchronises the device.  No reference counterpart (measurement only). */
int crag_index_phase_trace(crag_index *ix, uint64_t *out128);

/* Name of the scan kernel the most recent search on this index launched ("crag::scan_pipe_kernel", ...),
 * as rocprofv3 prints it; "" before the first search.  For bench.py's roofline object. */
const char *crag_index_last_scan_kernel(const crag_index *ix);

/* Bytes of one corpus row the prefilter scan streams: dim*2 when the index keeps the fp16 mirror of the unit rows
 * (default; + dim*2 bytes of HBM per row beside the dim*4 fp32 row, which stays the source of truth: candidates
 * are rescored from it), dim*4 when it was created with CRAG_NO_FP16_MIRROR=1, 0 with CRAG_NO_PREFILTER=1.
 * the bytes of a prefilter are declared separately -- bench.py's roofline uses this figure. */
int64_t crag_index_prefilter_row_bytes(const crag_index *ix);

/* Launch geometry of the scan kernel for the current size (for DESIGN/bench reporting). */
int crag_index_scan_geometry(const crag_index *ix, int nq, int *workgroups, int *threads,
                             int *query_blocks, int64_t *algorithmic_bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* CRAG_DENSE_H */
