/*
 * oracle/exact_scan.c — CPU restatement of the reference's dense exact-scan lane.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cadence_rag_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no test, fixture or
 * golden vector that pins any dense score or ordering (every DB-backed test
 * disables the dense lane: /root/reference/tests/conftest.py:85,95), and the
 * arithmetic lives in a third-party extension that is absent from
 * /root/reference: pgvector 0.8.1 (pinned at /root/reference/app/config.py:8,
 * docker-compose.yml:29).  This file restates
 *   (a) the reference's SQL call site, /root/reference/app/retrieve.py:339-353
 *       (`score = 1 - (embedding <=> q)`, `WHERE embedding IS NOT NULL`,
 *       `ORDER BY embedding <=> q LIMIT k`), and :369-388 for artifact_chunks;
 *   (b) pgvector 0.8.1's published cosine-distance algorithm
 *       (src/vector.c: VectorCosineSimilarity + cosine_distance): float
 *       accumulators for a.b, a.a, b.b over the dim in index order, then
 *       (double)dot / sqrt((double)na * (double)nb), clamped to [-1, 1],
 *       distance = 1.0 - similarity returned as float8.
 * The fixtures under tests/golden/ generated from this file are therefore the
 * build's own pin, not the reference's.
 *
 * Two arithmetic modes:
 *   CRAG_ORACLE_F32SEQ (0): (b) verbatim — fp32 sequential accumulation.
 *   CRAG_ORACLE_F64    (1): fp64 accumulation ("truth" for tolerance checks:
 *                           BASELINE.json asks |dscore| <= 1e-4 and same order).
 *
 * Ordering: ascending distance; the reference SQL has no tie-break
 * (retrieve.py:348), so we take the deterministic refinement "ascending id".
 * Rows whose similarity is NaN (zero-norm or non-finite rows: pgvector gives a
 * NaN distance that Postgres sorts last) are made ineligible, as is every row
 * whose mask bit is clear (the mask is the build's encoding of
 * _build_filter_clause, retrieve.py:93-120).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CRAG_ORACLE_F32SEQ 0
#define CRAG_ORACLE_F64 1

/* pgvector 0.8.1 src/vector.c VectorCosineSimilarity (restated). */
static double cosine_similarity_f32seq(int dim, const float *ax, const float *bx)
{
    float similarity = 0.0f;
    float norma = 0.0f;
    float normb = 0.0f;
    for (int i = 0; i < dim; i++) {
        similarity += ax[i] * bx[i];
        norma += ax[i] * ax[i];
        normb += bx[i] * bx[i];
    }
    /* "Use sqrt(a * b) over sqrt(a) * sqrt(b)" */
    return (double)similarity / sqrt((double)norma * (double)normb);
}

static double cosine_similarity_f64(int dim, const float *ax, const float *bx)
{
    double similarity = 0.0, norma = 0.0, normb = 0.0;
    for (int i = 0; i < dim; i++) {
        similarity += (double)ax[i] * (double)bx[i];
        norma += (double)ax[i] * (double)ax[i];
        normb += (double)bx[i] * (double)bx[i];
    }
    return similarity / sqrt(norma * normb);
}

/* pgvector cosine_distance: clamp similarity to [-1, 1], return 1 - similarity. */
static double cosine_distance_from_similarity(double similarity)
{
    if (similarity > 1.0)
        similarity = 1.0;
    else if (similarity < -1.0)
        similarity = -1.0;
    return 1.0 - similarity;
}

double crag_oracle_cosine_distance(int mode, int dim, const float *a, const float *b)
{
    double s = (mode == CRAG_ORACLE_F64) ? cosine_similarity_f64(dim, a, b)
                                         : cosine_similarity_f32seq(dim, a, b);
    if (isnan(s))
        return NAN;
    return cosine_distance_from_similarity(s);
}

typedef struct {
    double dist;
    int64_t id;
} cand_t;

/* "worse" = sorts later: larger distance, or equal distance and larger id. */
static inline int cand_worse(const cand_t *a, const cand_t *b)
{
    if (a->dist != b->dist)
        return a->dist > b->dist;
    return a->id > b->id;
}

/* bounded max-heap on "worse" (root = worst kept candidate), like a top-N heapsort */
static void heap_sift_down(cand_t *h, int n, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cand_worse(&h[l], &h[m]))
            m = l;
        if (r < n && cand_worse(&h[r], &h[m]))
            m = r;
        if (m == i)
            return;
        cand_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static void heap_sift_up(cand_t *h, int i)
{
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!cand_worse(&h[i], &h[p]))
            return;
        cand_t t = h[i];
        h[i] = h[p];
        h[p] = t;
        i = p;
    }
}

static int cand_cmp_best_first(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (cand_worse(a, b))
        return 1;
    if (cand_worse(b, a))
        return -1;
    return 0;
}

/*
 * Exact top-k for nq queries over n corpus rows (row-major fp32, `dim` wide).
 *   ids         nullable; NULL => id of row i is i
 *   mask        nullable; bit (i & 7) of byte mask[qi*mask_stride + (i >> 3)] set
 *               => row i eligible for query qi; mask_stride 0 => one shared mask
 *   out_ids     [nq, k]  -1 padded
 *   out_scores  [nq, k]  1 - distance (float8 arithmetic, as the SQL does), NaN padded
 *   out_counts  [nq]     number of valid entries (<= k)
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
int crag_oracle_topk(int mode, const float *queries, int nq, const float *corpus,
                     int64_t n, int dim, const int64_t *ids, const uint8_t *mask,
                     int64_t mask_stride, int k, int64_t *out_ids, double *out_scores,
                     int32_t *out_counts)
{
    if (nq < 0 || n < 0 || dim <= 0 || k <= 0)
        return -1;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int qi = 0; qi < nq; qi++) {
        cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
        if (!heap) {
            rc = -1;
            continue;
        }
        int hn = 0;
        const float *q = queries + (size_t)qi * dim;
        const uint8_t *m = mask ? mask + (size_t)qi * (size_t)mask_stride : NULL;
        for (int64_t i = 0; i < n; i++) {
            if (m && !((m[i >> 3] >> (i & 7)) & 1))
                continue;
            const float *c = corpus + (size_t)i * dim;
            double s = (mode == CRAG_ORACLE_F64) ? cosine_similarity_f64(dim, q, c)
                                                 : cosine_similarity_f32seq(dim, q, c);
            if (isnan(s))
                continue; /* zero-norm / non-finite row: ineligible (see header) */
            cand_t cd;
            cd.dist = cosine_distance_from_similarity(s);
            cd.id = ids ? ids[i] : i;
            if (hn < k) {
                heap[hn] = cd;
                heap_sift_up(heap, hn);
                hn++;
            } else if (cand_worse(&heap[0], &cd)) {
                heap[0] = cd;
                heap_sift_down(heap, hn, 0);
            }
        }
        qsort(heap, (size_t)hn, sizeof(cand_t), cand_cmp_best_first);
        for (int j = 0; j < k; j++) {
            if (j < hn) {
                out_ids[(size_t)qi * k + j] = heap[j].id;
                out_scores[(size_t)qi * k + j] = 1.0 - heap[j].dist;
            } else {
                out_ids[(size_t)qi * k + j] = -1;
                out_scores[(size_t)qi * k + j] = NAN;
            }
        }
        out_counts[qi] = hn;
        free(heap);
    }
    return rc;
}

/* All-pairs score matrix (1 - distance), NaN where ineligible; for small cases. */
int crag_oracle_scores(int mode, const float *queries, int nq, const float *corpus,
                       int64_t n, int dim, double *out /* [nq, n] */)
{
    if (nq < 0 || n < 0 || dim <= 0)
        return -1;
    for (int qi = 0; qi < nq; qi++)
        for (int64_t i = 0; i < n; i++) {
            double d = crag_oracle_cosine_distance(mode, dim, queries + (size_t)qi * dim,
                                                   corpus + (size_t)i * dim);
            out[(size_t)qi * n + i] = isnan(d) ? NAN : 1.0 - d;
        }
    return 0;
}

#ifdef _OPENMP
extern int omp_get_max_threads(void);
extern void omp_set_num_threads(int);
#endif

int crag_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void crag_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}
