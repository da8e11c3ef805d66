// BM25 channel: Okapi scoring over term-major CSR postings + fused top-k.
//
// Replaces `BM25Okapi.get_scores(tokens)` and the full Python sort at
// legalrag/retrieval/bm25_retriever.py:74-75.  One block owns (query, slab of
// documents): the slab's fp64 score vector lives in LDS, the query's tokens are
// walked IN QUERY ORDER (duplicates included) and each token's posting list is
// scattered into it by all threads — a posting list holds a document at most
// once, so there are no write conflicts and every document receives its
// contributions in exactly the order rank_bm25's `score += ...` loop applies
// them.  fp64, compiled with -ffp-contract=off: bit-identical scores.  The
// slab is then ranked in place (score desc, ties -> lower doc id, zero-score
// docs included) with the wave-level selector of topk.hpp.
// Algorithmic bytes per query: sum_t df(t)*12 (posting doc id + precomputed fp64 factor).
#include "common.hpp"
#include "topk.hpp"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>


#include "bm25_core.hpp"

using namespace amdr;

struct amdr_bm25 {
  int device = 0;
  int64_t n_terms = 0, n_docs = 0, nnz = 0;
  double avgdl = 0, k1 = 1.5, b = 0.75;
  long long* term_ptr = nullptr;
  int* post_doc = nullptr;
  double* post_w = nullptr;  // tf*(k1+1) / (tf + k1*(1 - b + b*len/avgdl)) per posting, fp64
  double* idf = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  DevBuf part[2], qterms, qptr, sbuf, ibuf, full;  // part[0]: "_device" calls, part[1]: host-pointer calls (see dense.hip)
  DevBuf ticket;  // 64 zeroed ints: arrival counters of the one-launch serving step (fuse.hip), self-resetting
};

namespace {

constexpr int kSlabMax = 4096;  // docs per block: 32 KiB of fp64 scores in LDS

struct BmPlan {
  int slab, nslabs, cap, cap_merge, waves;
  size_t lds, part_bytes;
};

void bm_plan(int64_t n_docs, int nq, int k, BmPlan* p) {
  // One wave per (query, slab) in every shape (measured on 1 260 ... 9 000 documents, 8 192
  // queries: the 4-wave block with its per-token block barriers and list combine lost to one
  // wave at every size — e.g. 1 260 docs, k = 10: 237 -> 74 us).  Shallow k (bm_use_argmax):
  // slabs of <= 2 048 documents ranked by the register arg-max (<= 32 scores per lane), its k
  // winners parked in a short list; otherwise slabs of <= 4 096 ranked by the staged selector.
  // Slabs are balanced (3 000 documents = 2 x 1 536, not 2 048 + 952) and merged by
  // bm25_merge_kernel.
  const int64_t n = n_docs > 0 ? n_docs : 1;
  auto balanced = [&](int slab_max) {
    p->nslabs = (int)((n + slab_max - 1) / slab_max);
    p->slab = (int)(((n + p->nslabs - 1) / p->nslabs + 15) / 16 * 16);
    if (p->slab > slab_max) p->slab = slab_max;
    p->nslabs = (int)((n + p->slab - 1) / p->slab);
  };
  balanced(kBmOneWaveDocs);
  const bool argmax = bm_use_argmax(k, p->slab);
  if (!argmax) balanced(kSlabMax);
  p->waves = 1;
  p->cap_merge = topk_cap(k);
  p->cap = argmax ? (k <= 16 ? 16 : kBmArgmaxK) : p->cap_merge;
  p->lds = (size_t)p->slab * sizeof(double) + (size_t)p->waves * p->cap * sizeof(C64) + 4 * sizeof(int) +
           128 * sizeof(long) + 8;  // token table (3 x kBmTok x 8 B) / ranking scratch (2 x 64 x 8 B)
  p->part_bytes = (size_t)p->nslabs * nq * k * sizeof(C64);
}

// AMDR_BM25_SELECT=0 pins the exact arg-max rounds (A/B and tests of the fallback path)
static bool bm_select_enabled() {
  const char* e = getenv("AMDR_BM25_SELECT");
  return !(e && e[0] == '0');
}

int bm_run(amdr_bm25* h, int ws, const int* q_terms_dev, const long long* q_ptr_dev, int nq, int k, double* scores_dev,
           int64_t* ids_dev, double* full_dev, hipStream_t st) {
  BmPlan p;
  bm_plan(h->n_docs, nq, k, &p);
  C64* part = nullptr;
  if (scores_dev) {
    int rc = h->part[ws].ensure(p.part_bytes);
    if (rc) return rc;
    part = h->part[ws].as<C64>();
  }
  const bool direct = scores_dev && p.nslabs == 1;  // single slab: the block's list is the answer
  double* fs = direct ? scores_dev : nullptr;
  long long* fi = direct ? (long long*)ids_dev : nullptr;
  if (direct) part = nullptr;
  if (p.lds > 48 * 1024) {  // never with the slab limits of bm_plan (4 096 x 8 B + lists < 48 KiB); guard for edits
    return fail(AMDR_EINVAL, "bm25: slab needs %zu B of LDS", p.lds);
  }
#define AMDR_BM_LAUNCH(NVT)                                                                                          \
  hipLaunchKernelGGL((bm25_score_topk_kernel<1, NVT>), dim3(p.nslabs, nq), dim3(64), p.lds, st, h->term_ptr,         \
                     h->post_doc, h->post_w, h->idf, (long)h->n_terms, (long)h->n_docs, q_terms_dev, q_ptr_dev, nq, k, \
                     p.cap, p.slab, bm_select_enabled() ? 1 : 0, full_dev, part, fs, fi)
  const int nv = bm_use_argmax(k, p.slab) ? (p.slab + 63) / 64 : 1;  // the staged selector needs no register bucket
  if (nv <= 4) AMDR_BM_LAUNCH(4);
  else if (nv <= 8) AMDR_BM_LAUNCH(8);
  else if (nv <= 10) AMDR_BM_LAUNCH(10);
  else if (nv <= 16) AMDR_BM_LAUNCH(16);
  else if (nv <= 20) AMDR_BM_LAUNCH(20);
  else AMDR_BM_LAUNCH(32);
#undef AMDR_BM_LAUNCH
  AMDR_HIP(hipGetLastError());
  if (scores_dev && !direct) {
    size_t lds = (size_t)kBmWaves * p.cap_merge * sizeof(C64) + kBmWaves * sizeof(int);
    hipLaunchKernelGGL(bm25_merge_kernel, dim3(nq), dim3(256), lds, st, part, p.nslabs, nq, k, p.cap_merge, scores_dev,
                       (long long*)ids_dev);
    AMDR_HIP(hipGetLastError());
  }
  return AMDR_OK;
}

}  // namespace
namespace amdr {
int bm25_small_raw(amdr_bm25_t* h, int nq, int k, Bm25Raw* out) {
  if (!h->ticket.p) return fail(AMDR_EINVAL, "bm25: handle without arrival counters");
  BmPlan p;
  bm_plan(h->n_docs, nq, k, &p);
  out->term_ptr = h->term_ptr;
  out->post_doc = h->post_doc;
  out->post_w = h->post_w;
  out->idf = h->idf;
  out->n_terms = (long)h->n_terms;
  out->n_docs = (long)h->n_docs;
  out->ticket = h->ticket.as<int>();
  out->slab = p.slab;
  out->nslabs = p.nslabs;
  out->cap = p.cap;
  out->lds = p.lds;
  out->argmax = bm_use_argmax(k, p.slab);
  out->nvt = out->argmax ? (p.slab + 63) / 64 : 1;
  out->select_on = bm_select_enabled();
  return AMDR_OK;
}
std::mutex& bm25_mutex(amdr_bm25_t* h) { return h->mu; }
}  // namespace amdr
namespace {

template <class T>
int upload(T** dst, const T* src, size_t count) {
  *dst = nullptr;
  size_t bytes = (count + 1) * sizeof(T);  // one padding element: clamped prefetch reads may touch [count]
  AMDR_HIP(hipMalloc((void**)dst, bytes));
  if (count) AMDR_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return AMDR_OK;
}

}  // namespace

extern "C" {

int amdr_bm25_create(const int64_t* term_ptr, const int32_t* post_doc, const int32_t* post_tf, const double* idf,
                     const int32_t* doc_len, int64_t n_terms, int64_t n_docs, double avgdl, double k1, double b,
                     int32_t device, amdr_bm25_t** out) {
  AMDR_REQUIRE(out != nullptr, "bm25_create: out is null");
  *out = nullptr;
  AMDR_REQUIRE(term_ptr && idf && doc_len, "bm25_create: null array");
  AMDR_REQUIRE(n_terms >= 0 && n_docs >= 1 && n_docs < (1ll << 31), "bm25_create: bad sizes");
  AMDR_REQUIRE(n_terms < (1ll << 31), "bm25_create: too many terms");
  AMDR_REQUIRE(term_ptr[0] == 0, "bm25_create: term_ptr[0] != 0");
  for (int64_t t = 0; t < n_terms; ++t)
    AMDR_REQUIRE(term_ptr[t + 1] >= term_ptr[t], "bm25_create: term_ptr not monotone at %lld", (long long)t);
  const int64_t nnz = term_ptr[n_terms];
  AMDR_REQUIRE(nnz == 0 || (post_doc && post_tf), "bm25_create: null postings");
  for (int64_t t = 0; t < n_terms; ++t)
    for (int64_t p = term_ptr[t]; p < term_ptr[t + 1]; ++p) {
      AMDR_REQUIRE(post_doc[p] >= 0 && post_doc[p] < n_docs, "bm25_create: posting %lld doc out of range", (long long)p);
      AMDR_REQUIRE(p == term_ptr[t] || post_doc[p] > post_doc[p - 1],
                   "bm25_create: postings of term %lld not strictly ascending", (long long)t);
    }
  int rc = check_device(device);
  if (rc) return rc;
  amdr_bm25* h = new (std::nothrow) amdr_bm25();
  if (!h) return fail(AMDR_ENOMEM, "bm25_create: host alloc");
  h->device = device;
  h->n_terms = n_terms;
  h->n_docs = n_docs;
  h->nnz = nnz;
  h->avgdl = avgdl;
  h->k1 = k1;
  h->b = b;
  // Per-posting factor of rank_bm25's expression, operand for operand in numpy's order
  // (this translation unit is built with -ffp-contract=off; IEEE fp64 mul/add/div are
  // correctly rounded on the host as on the device, so the value is the one numpy computes).
  std::vector<double> pw((size_t)nnz);
  {
    const double k1p1 = k1 + 1, one_minus_b = 1 - b;
    for (int64_t p = 0; p < nnz; ++p) {
      const double qf = (double)post_tf[p];
      const double dl = (double)doc_len[post_doc[p]];
      const double denom = qf + k1 * (one_minus_b + b * dl / avgdl);
      pw[(size_t)p] = qf * k1p1 / denom;
    }
  }
  rc = upload(&h->term_ptr, (const long long*)term_ptr, (size_t)n_terms + 1);
  if (!rc) rc = upload(&h->post_doc, post_doc, (size_t)nnz);
  if (!rc) rc = upload(&h->post_w, pw.data(), (size_t)nnz);
  if (!rc) rc = upload(&h->idf, idf, (size_t)n_terms);
  if (!rc && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    rc = fail(AMDR_EHIP, "bm25_create: stream");
  if (!rc) rc = h->ticket.ensure(64 * sizeof(int));  // arrival counters of the one-launch serving step: zero between launches
  if (!rc && hipMemset(h->ticket.p, 0, 64 * sizeof(int)) != hipSuccess) rc = fail(AMDR_EHIP, "bm25_create: memset");
  if (rc) {
    amdr_bm25_destroy(h);
    return rc;
  }
  *out = h;
  return AMDR_OK;
}

int amdr_bm25_ndocs(const amdr_bm25_t* h, int64_t* n) {
  AMDR_REQUIRE(h && n, "bm25_ndocs: null");
  *n = h->n_docs;
  return AMDR_OK;
}

int amdr_bm25_reserve(amdr_bm25_t* h, int32_t nq_max, int32_t k_max, int64_t total_terms_max) {
  AMDR_REQUIRE(h != nullptr, "bm25_reserve: null handle");
  AMDR_REQUIRE(nq_max >= 1 && k_max >= 1 && k_max <= AMDR_MAX_K && total_terms_max >= 0, "bm25_reserve: bad sizes");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  BmPlan p;
  bm_plan(h->n_docs, nq_max, k_max, &p);
  int rc = h->part[0].ensure(p.part_bytes);
  if (!rc) rc = h->qterms.ensure((size_t)(total_terms_max + 1) * sizeof(int));
  if (!rc) rc = h->qptr.ensure((size_t)(nq_max + 1) * sizeof(long long));
  if (!rc) rc = h->sbuf.ensure((size_t)nq_max * k_max * sizeof(double));
  if (!rc) rc = h->ibuf.ensure((size_t)nq_max * k_max * sizeof(int64_t));
  return rc;
}

static int bm_check(const amdr_bm25* h, const void* qt, const void* qp, int nq, int k) {
  AMDR_REQUIRE(h != nullptr, "bm25: null handle");
  AMDR_REQUIRE(nq >= 0, "bm25: nq=%d", nq);
  AMDR_REQUIRE(k >= 1 && k <= AMDR_MAX_K, "bm25: k=%d outside [1,%d]", k, AMDR_MAX_K);
  AMDR_REQUIRE(nq == 0 || qp, "bm25: null q_ptr");
  (void)qt;
  return AMDR_OK;
}

int amdr_bm25_search_device(amdr_bm25_t* h, const int32_t* q_terms_dev, const int64_t* q_ptr_dev, int32_t nq,
                            int32_t k, double* scores_dev, int64_t* ids_dev, void* stream) {
  int rc = bm_check(h, q_terms_dev, q_ptr_dev, nq, k);
  if (rc) return rc;
  AMDR_REQUIRE(nq == 0 || (scores_dev && ids_dev), "bm25: null output");
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  return bm_run(h, 0, q_terms_dev, (const long long*)q_ptr_dev, nq, k, scores_dev, ids_dev, nullptr, (hipStream_t)stream);
}

static int bm_stage_queries(amdr_bm25* h, const int32_t* q_terms, const int64_t* q_ptr, int nq) {
  AMDR_REQUIRE(q_ptr[0] == 0, "bm25: q_ptr[0] != 0");
  for (int i = 0; i < nq; ++i) AMDR_REQUIRE(q_ptr[i + 1] >= q_ptr[i], "bm25: q_ptr not monotone");
  const int64_t tot = q_ptr[nq];
  AMDR_REQUIRE(tot == 0 || q_terms, "bm25: null q_terms");
  int rc = h->qterms.ensure((size_t)(tot + 1) * sizeof(int));
  if (!rc) rc = h->qptr.ensure((size_t)(nq + 1) * sizeof(long long));
  if (rc) return rc;
  if (tot) AMDR_HIP(hipMemcpyAsync(h->qterms.p, q_terms, (size_t)tot * sizeof(int), hipMemcpyHostToDevice, h->stream));
  AMDR_HIP(hipMemcpyAsync(h->qptr.p, q_ptr, (size_t)(nq + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  return AMDR_OK;
}

int amdr_bm25_search(amdr_bm25_t* h, const int32_t* q_terms, const int64_t* q_ptr, int32_t nq, int32_t k,
                     double* scores_host, int64_t* ids_host) {
  int rc = bm_check(h, q_terms, q_ptr, nq, k);
  if (rc) return rc;
  AMDR_REQUIRE(nq == 0 || (scores_host && ids_host), "bm25: null output");
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  if ((rc = bm_stage_queries(h, q_terms, q_ptr, nq))) return rc;
  if ((rc = h->sbuf.ensure((size_t)nq * k * sizeof(double)))) return rc;
  if ((rc = h->ibuf.ensure((size_t)nq * k * sizeof(int64_t)))) return rc;
  rc = bm_run(h, 1, h->qterms.as<int>(), h->qptr.as<long long>(), nq, k, h->sbuf.as<double>(), h->ibuf.as<int64_t>(),
              nullptr, h->stream);
  if (rc) return rc;
  AMDR_HIP(hipMemcpyAsync(scores_host, h->sbuf.p, (size_t)nq * k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipMemcpyAsync(ids_host, h->ibuf.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipStreamSynchronize(h->stream));
  return AMDR_OK;
}

int amdr_bm25_scores(amdr_bm25_t* h, const int32_t* q_terms, const int64_t* q_ptr, int32_t nq, double* scores_host) {
  int rc = bm_check(h, q_terms, q_ptr, nq, 1);
  if (rc) return rc;
  AMDR_REQUIRE(nq == 0 || scores_host, "bm25_scores: null output");
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  if ((rc = bm_stage_queries(h, q_terms, q_ptr, nq))) return rc;
  if ((rc = h->full.ensure((size_t)nq * h->n_docs * sizeof(double)))) return rc;
  rc = bm_run(h, 1, h->qterms.as<int>(), h->qptr.as<long long>(), nq, 1, nullptr, nullptr, h->full.as<double>(),
              h->stream);
  if (rc) return rc;
  AMDR_HIP(hipMemcpyAsync(scores_host, h->full.p, (size_t)nq * h->n_docs * sizeof(double), hipMemcpyDeviceToHost,
                          h->stream));
  AMDR_HIP(hipStreamSynchronize(h->stream));
  return AMDR_OK;
}

int amdr_bm25_destroy(amdr_bm25_t* h) {
  if (!h) return AMDR_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamDestroy(h->stream);
  }
  if (h->term_ptr) (void)hipFree(h->term_ptr);
  if (h->post_doc) (void)hipFree(h->post_doc);
  if (h->post_w) (void)hipFree(h->post_w);
  if (h->idf) (void)hipFree(h->idf);
  h->part[0].release();
  h->part[1].release();
  h->qterms.release();
  h->qptr.release();
  h->sbuf.release();
  h->ibuf.release();
  h->full.release();
  h->ticket.release();
  delete h;
  return AMDR_OK;
}

}  // extern "C"
