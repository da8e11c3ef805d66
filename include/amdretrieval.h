/*
 * amdretrieval.h — C ABI of libamdretrieval.so (MI355X / gfx950 hybrid-retrieval kernels).
 *
 * The reference (Fan-Luo/Legal-RAG) is 100 % Python and has no FFI of its own
 * (SURVEY.md §8b); its native arithmetic lives in third-party wheels.  Each entry
 * point below therefore cites the reference CALL SITE whose native callee it
 * replaces.  Conventions: plain pointers and sizes only; every function returns
 * 0 on success or a negative AMDR_E* code, with a thread-local message available
 * from amdr_last_error(); the caller allocates all outputs; opaque handles are
 * safe for concurrent read-only searches (internally serialised per handle),
 * mutation (add/destroy) must not race with searches on the same handle.
 *
 * "_device" variants take DEVICE pointers and a hipStream_t passed as void*
 * (0 = the null stream); they enqueue work and return without synchronising,
 * and are graph-capturable once amdr_*_reserve() has sized the workspace
 * (reserve covers the "_device" workspace; the host-pointer calls size their own on first use).
 * A handle owns TWO workspaces: one for the "_device" calls, one for the plain
 * (host-pointer) calls.  "_device" calls on the same handle must be ordered with
 * respect to each other by the caller (same stream, or events between streams) —
 * the library only enqueues and cannot know when the caller's stream gets there.
 * Plain variants take HOST pointers, run on the handle's private stream inside
 * the handle's mutex and return after the results are in the host buffers: they
 * are safe to call from any number of threads, also while "_device" work of the
 * same handle is still in flight on another stream (different workspace).
 */
#ifndef AMDRETRIEVAL_H
#define AMDRETRIEVAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMDR_OK 0
#define AMDR_EINVAL (-1)   /* bad argument (null, shape, k out of range) */
#define AMDR_EHIP (-2)     /* a HIP runtime call failed; see amdr_last_error() */
#define AMDR_ENOMEM (-3)   /* device or host allocation failed */
#define AMDR_ENODEV (-4)   /* no usable gfx950 device */

#define AMDR_MAX_K 256      /* per-channel depth limit of the fused top-k kernels */
#define AMDR_MAX_DIM 1024   /* dense embedding dim limit (multiple of 4) */
#define AMDR_MAXSIM_DIM 128 /* ColBERT token dim (fixed by the kernel's register layout) */
#define AMDR_MAXSIM_QLEN 32 /* ColBERT query length after [MASK] padding */

typedef struct amdr_dense amdr_dense_t;
typedef struct amdr_dense_small amdr_dense_small_t;
typedef struct amdr_bm25 amdr_bm25_t;
typedef struct amdr_maxsim amdr_maxsim_t;
typedef struct amdr_tokenizer amdr_tokenizer_t;

/* ---- library ---------------------------------------------------------- */
const char* amdr_last_error(void);
int amdr_version(void);                     /* 10000*major + 100*minor + patch */
int amdr_device_count(int32_t* count);      /* number of visible HIP devices   */
int amdr_device_name(int32_t device, char* buf, int32_t buf_len); /* gcnArchName */

/* ---- dense channel: exact inner-product top-k -------------------------
 * Replaces faiss `index.add(emb)` (legalrag/retrieval/builders/faiss_builder.py:91,
 * incremental_dense_builder.py:62) and `index.search(q_vec, k)`
 * (legalrag/retrieval/dense_retriever.py:42, vector_store.py:169).
 * X: row-major fp32 [n, d] (rows L2-normalised by the encoder).  Results are
 * sorted by score descending, ties -> lower row id, padded with id -1 and
 * score -FLT_MAX when k > n (faiss convention). */
int amdr_dense_create(const float* X_host, int64_t n, int32_t d, int32_t device, amdr_dense_t** out);
/* adopt an existing device matrix without copying (the caller keeps ownership
 * and must keep it alive); used for shards generated in HBM.  The matrix is READ at creation (largest component and
 * row norm, for the fp16 first pass of large scans): the call synchronises the device first, so work that fills X on
 * any stream and was enqueued before the call is complete; the matrix must not change while the handle lives. */
int amdr_dense_create_from_device(const float* X_dev, int64_t n, int32_t d, int32_t device, amdr_dense_t** out);
int amdr_dense_add(amdr_dense_t* h, const float* X_host, int64_t n_add);
int amdr_dense_ntotal(const amdr_dense_t* h, int64_t* n);
int amdr_dense_dim(const amdr_dense_t* h, int32_t* d);
int amdr_dense_reserve(amdr_dense_t* h, int32_t nq_max, int32_t k_max);
int amdr_dense_search(amdr_dense_t* h, const float* Q_host, int32_t nq, int32_t k,
                      float* scores_host, int64_t* ids_host);
int amdr_dense_search_device(amdr_dense_t* h, const float* Q_dev, int32_t nq, int32_t k,
                             float* scores_dev, int64_t* ids_dev, void* stream);
/* scores of an explicit candidate list: out[q, j] = <Q[q], X[rows[q, j]]>, -FLT_MAX for
 * rows outside [0, n).  Building block for GraphRetriever's candidate rescoring
 * (legalrag/retrieval/graph_retriever.py:177-191 re-embeds up to graph_limit=800 candidate
 * texts per query and takes cosines; the embeddings are already resident here). */
int amdr_dense_score_rows(amdr_dense_t* h, const float* Q_host, int32_t nq, const int64_t* rows_host, int32_t m,
                          float* scores_host);
/* copy rows [row0, row0+nrows) back to the host (used by parity tests to run
 * the oracle on exactly the matrix that is resident in HBM) */
int amdr_dense_read_rows(const amdr_dense_t* h, int64_t row0, int64_t nrows, float* out_host);
/* Which kernels a search of nq queries at depth k would launch on this index, and how the work
 * is cut (e.g. "dense_panel_scores_kernel nb=6 parts=7 blocks=2044 + scores_slab_topk_kernel"):
 * written NUL-terminated into buf.  No device work.  bench.py names its roofline kernel with it. */
int amdr_dense_plan_info(const amdr_dense_t* h, int32_t nq, int32_t k, char* buf, int32_t buf_len);
/* Host-only (no device is touched): workspace bytes a batched search of nq queries at depth k on an [n, d] matrix
 * RESERVES before its passes — out6[0..2] = score / tile-maxima matrix, slab lists, candidate-tile lists — and the
 * maximum any single pass (full chunks and the remainder) then USES — out6[3..5].  Zeros for the 1-4 query forms.
 * A test holds out6[3+i] <= out6[i] over shapes with many row slabs (tests/test_abi.py). */
int amdr_dense_workspace_plan(int64_t n, int32_t d, int32_t nq, int32_t k, int64_t* out6);
/* Large scans (chunk matrix far beyond the caches, >= 5 queries): the first pass of the two-level top-k runs on the
 * fp16 matrix instructions over fp16 roundings of both operands, 64 queries per scan; its candidate cut is widened by a
 * proven rounding bound and the second pass is the exact fp32 kernel, so ids and score bits are those of the exact
 * forms (csrc/dense_hi.hip).  A query whose cut the bound does not separate sends its batch through the exact first
 * pass as well (decided on the device).  The width of the candidate cut adapts per handle: k + max(k, 22 / 54 / 96)
 * + 1 tiles per query (levels 0-2).  One unresolved query sends its whole pass (<= 64 queries) through the exact chain,
 * so passes are what is counted: more than 10 % of >= 4 passes flagged moves the handle up a level, at the top level
 * more than half make it give the fp16 pass up (exact passes from then on); AMDR_DENSE_HI_LEVEL pins the level.
 * out6[0] = queries that took the fp16 first pass since creation, out6[1] = those it could not resolve, out6[2] =
 * current level, out6[3] = 1 while the pass is in use, out6[4] = passes, out6[5] = passes whose flag went up.
 * Synchronises the device.  The matrix wrapped by amdr_dense_create_from_device must not change while the handle
 * lives (its largest component and row norm are measured at creation). */
int amdr_dense_hi_counters(amdr_dense_t* h, int64_t* out6);
/* HIP-event bracket around the scan kernel alone (not the merge), recorded on
 * the stream each search is launched on; used by bench.py for the roofline.
 * begin() arms up to max_launches event pairs, end() returns the summed scan
 * time and the number of launches measured since begin(). */
int amdr_dense_profile_begin(amdr_dense_t* h, int32_t max_launches);
int amdr_dense_profile_end(amdr_dense_t* h, double* total_ms, int32_t* launches);
int amdr_dense_destroy(amdr_dense_t* h);

/* ---- BM25 channel: Okapi scoring over term-major CSR postings ----------
 * Replaces `BM25Okapi.get_scores(tokens)` + the full Python sort
 * (legalrag/retrieval/bm25_retriever.py:74-75).  float64 throughout, no FMA
 * contraction; per document the query tokens are accumulated in query order
 * with duplicates counted, so scores are bit-identical to rank_bm25's numpy
 * expression.  Ties -> lower doc id; zero-score docs ARE returned.
 * term_ptr[n_terms+1] indexes post_doc/post_tf (ascending doc id per term);
 * idf already has rank_bm25's epsilon floor applied. */
int amdr_bm25_create(const int64_t* term_ptr, const int32_t* post_doc, const int32_t* post_tf,
                     const double* idf, const int32_t* doc_len, int64_t n_terms, int64_t n_docs,
                     double avgdl, double k1, double b, int32_t device, amdr_bm25_t** out);
int amdr_bm25_ndocs(const amdr_bm25_t* h, int64_t* n);
int amdr_bm25_reserve(amdr_bm25_t* h, int32_t nq_max, int32_t k_max, int64_t total_terms_max);
/* queries as CSR: q_terms[q_ptr[i] .. q_ptr[i+1]) = term ids of query i in
 * token order (unknown tokens: any negative id, they score 0) */
int amdr_bm25_search(amdr_bm25_t* h, const int32_t* q_terms, const int64_t* q_ptr, int32_t nq, int32_t k,
                     double* scores_host, int64_t* ids_host);
int amdr_bm25_search_device(amdr_bm25_t* h, const int32_t* q_terms_dev, const int64_t* q_ptr_dev,
                            int32_t nq, int32_t k, double* scores_dev, int64_t* ids_dev, void* stream);
/* dense score vectors [nq, n_docs] == BM25Okapi.get_scores per query */
int amdr_bm25_scores(amdr_bm25_t* h, const int32_t* q_terms, const int64_t* q_ptr, int32_t nq,
                     double* scores_host);
int amdr_bm25_destroy(amdr_bm25_t* h);

/* ---- BM25 query tokeniser + vocabulary lookup, batched (host code) -------
 * Replaces, per query, `tokens = list(jieba.cut(query))` and the per-token vocabulary lookup of
 * `BM25Okapi.get_scores` (legalrag/retrieval/bm25_retriever.py:73-74) for text WITHOUT Han characters — the
 * case jieba's default mode decides without its dictionary (rule: legal-rag_amd/text.py, which stays the executable
 * specification; csrc/tokenize.cpp).  Queries are not lower-cased (the reference does not).  A query holding a Han
 * character is not tokenised: needs_segmenter[q] = 1, it gets no terms, and the caller takes its own segmenter.
 * vocab: n_terms UTF-8 strings, term i = vocab_blob[vocab_offsets[i] .. vocab_offsets[i+1]).
 * encode: queries as one UTF-8 blob + offsets [nq+1]; writes the CSR amdr_bm25_search takes — term_ids (capacity:
 * the blob's byte length always suffices; unknown token = -1) and q_ptr [nq+1].  No device work; thread-safe. */
int amdr_tokenizer_create(const char* vocab_blob, const int64_t* vocab_offsets, int64_t n_terms, amdr_tokenizer_t** out);
int amdr_tokenizer_encode(const amdr_tokenizer_t* t, const char* text_blob, const int64_t* text_offsets, int32_t nq,
                          int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter);
/* the same for queries joined by ONE NUL byte each (nq - 1 separators in n_bytes; a caller with a list of Python strings
 * builds this blob with two C-level operations — "\0".join(qs).encode() — instead of one encode per query).  Both forms
 * cut a batch into ranges of queries for a small persistent pool of worker threads (AMDR_TOKENIZER_THREADS, default: the
 * hardware's, at most 32; one worker per 256 queries) and splice the ranges' terms by a prefix sum. */
int amdr_tokenizer_encode_joined(const amdr_tokenizer_t* t, const char* text_blob, int64_t n_bytes, int32_t nq,
                                 int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter);
/* the same for queries that lie where they are: texts[q] = the n_bytes[q] UTF-8 bytes of query q (no blob is built; a
 * Python caller takes the pointers from the str objects themselves, csrc/pystrings.c) */
int amdr_tokenizer_encode_ptrs(const amdr_tokenizer_t* t, const char* const* texts, const int64_t* n_bytes, int32_t nq,
                               int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter);
/* byte spans of one text's tokens (tests compare them with text.jieba_cut); *n_tokens = -1: Han text */
int amdr_tokenizer_spans(const char* text, int64_t n_bytes, int32_t* starts, int32_t* ends, int32_t capacity,
                         int32_t* n_tokens);
int amdr_tokenizer_destroy(amdr_tokenizer_t* t);

/* ---- ColBERT channel: exhaustive late-interaction MaxSim ---------------
 * Replaces `Searcher.search(query, k)` (legalrag/retrieval/colbert_retriever.py:152).
 * D: fp32 token embeddings [total_tokens, 128], doc_ptr[n_docs+1] (every doc
 * has >= 1 token); Q: [nq, 32, 128].  score = sum_i max_j <q_i, d_j>.
 * Arithmetic: every operand is split exactly into fp16 hi + lo / 2048 (22 significant bits) and a product is
 * hi*hi + (hi*lo + lo*hi) / 2048 on the fp16 matrix instructions with fp32 accumulation — scores within ~3e-6 of
 * the fp64 definition on unit-norm tokens (north_star's bar: 1e-4), scale-free for any finite input; a store
 * holding a NaN / infinity, or AMDR_MAXSIM_F16X3=0, takes the exact fp32-input matrix form (csrc/maxsim.hip). */
int amdr_maxsim_create(const float* D_host, const int64_t* doc_ptr, int64_t n_docs, int32_t dim,
                       int32_t device, amdr_maxsim_t** out);
int amdr_maxsim_ndocs(const amdr_maxsim_t* h, int64_t* n);
/* which kernels a search of nq queries would launch and in which arithmetic form (NUL-terminated; no device work) */
int amdr_maxsim_plan_info(const amdr_maxsim_t* h, int32_t nq, char* buf, int32_t buf_len);
int amdr_maxsim_reserve(amdr_maxsim_t* h, int32_t nq_max, int32_t k_max);
int amdr_maxsim_search(amdr_maxsim_t* h, const float* Q_host, int32_t nq, int32_t q_len, int32_t k,
                       float* scores_host, int64_t* ids_host);
int amdr_maxsim_search_device(amdr_maxsim_t* h, const float* Q_dev, int32_t nq, int32_t q_len, int32_t k,
                              float* scores_dev, int64_t* ids_dev, void* stream);
int amdr_maxsim_scores(amdr_maxsim_t* h, const float* Q_host, int32_t nq, int32_t q_len, float* scores_host);
int amdr_maxsim_destroy(amdr_maxsim_t* h);

/* ---- fusion + rerank blend ---------------------------------------------
 * Replaces HybridRetriever._fuse (legalrag/retrieval/hybrid_retriever.py:389-551)
 * with its helpers _minmax (:24-30) and _rrf_with_breakdown (:33-56), the
 * min_final_score filter (:309-310) and the rerank blend (:338-355 with
 * rerankers.py:48-54,349).  float64, no FMA contraction: bit-identical to the
 * Python float arithmetic.  Exactly tied fused scores keep first-appearance
 * order (dense list, then bm25, then colbert). */
#define AMDR_FUSE_RRF_NORM_BLEND 0
#define AMDR_FUSE_RRF 1
#define AMDR_FUSE_WRRF 2
#define AMDR_FUSE_WEIGHTED_SUM 3

typedef struct amdr_fuse_params {
  int32_t method;          /* AMDR_FUSE_* */
  int32_t rrf_k;           /* cfg.retrieval.rrf_k (60) */
  double alpha;            /* cfg.retrieval.rrf_alpha (0.5) */
  double w_dense, w_bm25, w_colbert;
  double min_final_score;  /* drop fused hits below this; -inf keeps all */
} amdr_fuse_params_t;

/* values per fused candidate, AMDR_FUSE_NVALS doubles each */
#define AMDR_FUSE_NVALS 9
#define AMDR_FV_SCORE 0
#define AMDR_FV_RRF_NORM 1
#define AMDR_FV_WSUM 2
#define AMDR_FV_NORM_DENSE 3
#define AMDR_FV_NORM_BM25 4
#define AMDR_FV_NORM_COLBERT 5
#define AMDR_FV_CONTRIB_DENSE 6
#define AMDR_FV_CONTRIB_BM25 7
#define AMDR_FV_CONTRIB_COLBERT 8

/* Inputs per channel: ids [nq, k_c] (-1 = padding, valid entries first, sorted
 * by score descending), scores [nq, k_c]; a channel with k_c == 0 is absent.
 * *_row2uid (nullable) maps a channel's row id to the corpus-wide chunk uid.
 * Outputs, max_out = kd+kb+kc entries per query, in fused-rank order:
 * out_ids [nq,max_out], out_vals [nq,max_out,AMDR_FUSE_NVALS], out_mask
 * [nq,max_out] (bit0 dense, bit1 bm25, bit2 colbert membership), out_count
 * [nq] = hits surviving the min_final_score filter. */
/* host-pointer form: every channel's scores as double (callers of _fuse may
 * hand in arbitrary Python floats; fp32 channel scores widen exactly) */
int amdr_fuse(const amdr_fuse_params_t* p, int32_t nq,
              const int64_t* dense_ids, const double* dense_scores, int32_t kd,
              const int64_t* bm25_ids, const double* bm25_scores, int32_t kb,
              const int64_t* colbert_ids, const double* colbert_scores, int32_t kc,
              int64_t* out_ids, double* out_vals, int32_t* out_mask, int32_t* out_count);
int amdr_fuse_device(const amdr_fuse_params_t* p, int32_t nq,
                     const int64_t* dense_ids, const float* dense_scores, int32_t kd, const int64_t* dense_row2uid,
                     const int64_t* bm25_ids, const double* bm25_scores, int32_t kb, const int64_t* bm25_row2uid,
                     const int64_t* colbert_ids, const float* colbert_scores, int32_t kc, const int64_t* colbert_row2uid,
                     int64_t* out_ids, double* out_vals, int32_t* out_mask, int32_t* out_count,
                     int32_t device, void* stream);

/* amdr_dense_search_device followed by amdr_fuse_device(dense lists, BM25 lists, no ColBERT) as ONE call — the same
 * outputs, bit for bit (dense_scores / dense_ids: the dense channel's own top-k; out_*: the fusion's), fewer launches:
 * for the serving corpora under a batch (<= 1 024 rows, k + kb <= 32) the rows are ranked and fused by one kernel
 * (two queries per wave), every other shape runs the two launches inside.  Reference stages:
 * hybrid_retriever.py:181-189 (search_dense) + :389-551 (_fuse).  AMDR_DENSE_FUSE=0 pins the two launches. */
int amdr_dense_search_fuse_device(amdr_dense_t* h, const float* Q_dev, int32_t nq, int32_t k,
                                  const amdr_fuse_params_t* p, const int64_t* dense_row2uid,
                                  const int64_t* bm25_ids, const double* bm25_scores, int32_t kb,
                                  const int64_t* bm25_row2uid, float* dense_scores_dev, int64_t* dense_ids_dev,
                                  int64_t* out_ids, double* out_vals, int32_t* out_mask, int32_t* out_count,
                                  void* stream);

/* The serving call — HybridRetriever.search(), one query at a time (hybrid_retriever.py:282-384: search_dense :181-189,
 * search_bm25 :191-209, _fuse :389-551) — as ONE launch: amdr_bm25_search_device + amdr_dense_search_fuse_device, the same
 * five outputs bit for bit (bm25_scores/ids [nq,kb], dense_scores/ids [nq,kd], out_* as amdr_fuse_device), for 1-4
 * queries on a corpus of <= 2 048 chunks held by both indexes on the same device with kd + kb <= 32: blocks of the one
 * grid score the BM25 slab or 4-16 chunk rows each, and the last block of a query to arrive ranks and fuses.  Every other
 * shape (and AMDR_HYBRID_SMALL=0) runs the two calls inside.  Only enqueues; after amdr_dense_reserve + amdr_bm25_reserve
 * it allocates nothing (capturable).  One call at a time per pair of handles (they share the handles' workspaces). */
int amdr_hybrid_small_device(amdr_dense_t* dense, amdr_bm25_t* bm25, const float* Q_dev, const int32_t* q_terms_dev,
                             const int64_t* q_ptr_dev, int32_t nq, int32_t kd, int32_t kb, const amdr_fuse_params_t* p,
                             const int64_t* dense_row2uid, const int64_t* bm25_row2uid, float* dense_scores_dev,
                             int64_t* dense_ids_dev, double* bm25_scores_dev, int64_t* bm25_ids_dev, int64_t* out_ids,
                             double* out_vals, int32_t* out_mask, int32_t* out_count, void* stream);

/* The first pass of the two-pass form of the long-batch dense channel on a short corpus (search_dense over a batch,
 * hybrid_retriever.py:181-189) — what amdr_dense_search_device / amdr_dense_search_fuse_device run inside from 4 096
 * queries per launch on <= 1 024 rows with d a multiple of 128 and k <= 12 (AMDR_DENSE_SMALL_HI=0: the
 * exact fp32 form), exported for tests and measurements.  approx_device writes, for every
 * query and chunk row, the dot product of the fp16 roundings of the scaled operands — S[nq, ldS] (ldS >= the rows padded to
 * 32, a multiple of 4; columns [n, padded) are 0) — on the fp16 matrix instructions, and per query the PROVEN bound
 * eps[q] >= |S[q][r] - <Q[q], X[r]>| for every row r (NaN: a non-finite query or one outside the scale range — no bound),
 * in the units of the exact score.  The second pass (inside the search calls) re-scores the rows with
 * S >= (k-th best of S) - 2 eps in exact fp32 and ranks them: the exact top-k, ties to the lower row.  create takes the statistics and the fp16 image of the dense handle's matrix as it is NOW (rows added
 * later are not seen); d must be a multiple of 128 in [128, 1024]; the dense handle must outlive this one. */
/* queries of this handle's two-pass searches so far that re-scored their WHOLE row inside the second pass (more than 32 rows
 * inside the margin, or no bound for the query): the results are exact either way, a large share means the data does not suit
 * the form (AMDR_DENSE_SMALL_HI=0).  Synchronises with the device. */
int amdr_dense_two_pass_fallbacks(amdr_dense_t* h, int64_t* out);
int amdr_dense_small_create(amdr_dense_t* dense, amdr_dense_small_t** out);
int amdr_dense_small_approx_device(amdr_dense_small_t* h, const float* Q_dev, int32_t nq, float* S_dev, int64_t ldS,
                                   float* eps_dev /* nullable [nq] */, void* stream);
int amdr_dense_small_destroy(amdr_dense_small_t* h);

/* Rerank blend over the first min(top_n, count[q]) fused hits of each query:
 * norm = minmax(ce_raw); score = (1-beta)*score + beta*norm; the candidates
 * are re-ordered by norm (stable), written back in front of the rest, and the
 * whole list is stably re-sorted by the new score.  In place on ids/vals/mask;
 * out_rerank [nq,max_out,2] receives (raw, norm) per OUTPUT position, NaN for
 * hits that were not reranked. ce_raw: [nq, top_n]. */
int amdr_rerank_blend(int32_t nq, int32_t max_out, const int32_t* count, int64_t* ids, double* vals,
                      int32_t* mask, const double* ce_raw, int32_t top_n, double beta, double* out_rerank);
int amdr_rerank_blend_device(int32_t nq, int32_t max_out, const int32_t* count, int64_t* ids, double* vals,
                             int32_t* mask, const double* ce_raw, int32_t top_n, double beta,
                             double* out_rerank, int32_t device, void* stream);

/* The columns a bulk caller reads from a fused result, compacted on the device: the first w fused hits of every query
 * as out_rows / out_scores / out_mask [nq, w] (entries past min(count[q], w): -1 / 0.0 / 0) and out_count [nq] =
 * min(count[q], w).  Inputs: the outputs of amdr_fuse_device (after the optional rerank blend).  One small device-to-host
 * copy then serves HybridRetriever.search_batch_arrays (hybrid_retriever.py:309-310 + :384, the cut to top_k). */
int amdr_fuse_compact_device(int32_t nq, int32_t max_out, int32_t w, const int64_t* ids, const double* vals,
                             const int32_t* mask, const int32_t* count, int64_t* out_rows, double* out_scores,
                             int32_t* out_mask, int32_t* out_count, int32_t device, void* stream);

/* ---- multi-GPU: merge per-shard top-k after the RCCL all-gather --------
 * No reference counterpart (the reference is single-process).
 * parts: [n_parts, nq, k_in] scores + GLOBAL ids (-1 padding); output
 * [nq, k_out], score descending, ties -> lower global id. */
int amdr_merge_topk_f32_device(const float* scores, const int64_t* ids, int32_t n_parts, int32_t nq, int32_t k_in,
                               int32_t k_out, float* out_scores, int64_t* out_ids, int32_t device, void* stream);
int amdr_merge_topk_f64_device(const double* scores, const int64_t* ids, int32_t n_parts, int32_t nq, int32_t k_in,
                               int32_t k_out, double* out_scores, int64_t* out_ids, int32_t device, void* stream);

/* ---- multi-GPU: the shard exchange of ALL channels, one launch either side of the all-gather -------------------
 * No reference counterpart (single process; legalrag/config.py:106 `colbert_nranks = 1`).  SURVEY.md 8(b)/8(e):
 * rank r holds rows [offset_r, offset_r + n_r) of every channel; per query batch ONE all-gather of a packed int64
 * buffer carries every channel's per-shard top-k.
 *   row of query q (amdr_shard_row_words words) = for each channel c, in order:
 *       k_c score words (the score's bits as fp64; an fp32 score widens exactly) | k_c GLOBAL ids (-1 = padding)
 * amdr_shard_pack_device : this rank's lists (scores fp32 or fp64 [nq, k_c], LOCAL ids [nq, k_c]) -> send [nq, row];
 *                          global id = local id + id_offset.
 * amdr_shard_merge_device: gathered [world, nq, row] (the all-gather's output, read in place) -> per channel the global
 *                          top-k_c: out scores [nq, k_c] (fp32 / fp64 as flagged), out ids [nq, k_c]; score descending,
 *                          ties -> lower global id, -1 / -FLT_MAX (-DBL_MAX) padding.  Identical on every rank.
 * Both only enqueue on `stream` (graph-capturable); up to 4 channels, k_c <= AMDR_MAX_K. */
typedef struct amdr_shard_chan {
  void* scores;   /* pack: const input lists; merge: output */
  int64_t* ids;
  int32_t k;
  int32_t f64;    /* 1: scores are double (BM25), 0: float (dense, MaxSim) */
} amdr_shard_chan_t;
int amdr_shard_row_words(const amdr_shard_chan_t* chans, int32_t n_chan, int64_t* words);
int amdr_shard_pack_device(const amdr_shard_chan_t* chans, int32_t n_chan, int32_t nq, int64_t id_offset, int64_t* send,
                           int32_t device, void* stream);
int amdr_shard_merge_device(const int64_t* gathered, int32_t world, int32_t nq, const amdr_shard_chan_t* out_chans,
                            int32_t n_chan, int32_t device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMDRETRIEVAL_H */
