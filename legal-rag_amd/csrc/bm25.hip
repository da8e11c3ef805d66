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
#include <mutex>
#include <new>
#include <vector>

namespace amdr {

constexpr int kBmWaves = 4;

__device__ __forceinline__ long lower_bound_i32(const int* __restrict__ a, long lo, long hi, int key) {
  while (lo < hi) {
    long mid = (lo + hi) >> 1;
    if (a[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// grid: (x = doc slabs, y = queries).  LDS: double sc[slab] + C64 lists[WAVES][cap] + int cnts[4] + token table [64]
// WAVES = 1 for small slabs (one wave per query: no block barriers, no list combine),
// 4 for full 4096-doc slabs.  With a single slab the final (scores, ids) are written
// directly and the merge launch is skipped.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bm25_score_topk_kernel(
    const long long* __restrict__ term_ptr, const int* __restrict__ post_doc, const double* __restrict__ post_w,
    const double* __restrict__ idf, long n_terms, long n_docs, const int* __restrict__ q_terms,
    const long long* __restrict__ q_ptr, int nq, int k, int cap, int slab,
    double* __restrict__ scores_out /* nullable [nq, n_docs] */, C64* __restrict__ part /* nullable [nslabs][nq][k] */,
    double* __restrict__ fin_scores /* nullable [nq,k]: single slab */, long long* __restrict__ fin_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* sc = reinterpret_cast<double*>(smem);
  C64* lists = reinterpret_cast<C64*>(sc + slab);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)WAVES * cap);
  long* tk_ps = reinterpret_cast<long*>(cnts + 4);  // [64] posting range + idf of up to 64 query tokens
  long* tk_pe = tk_ps + 64;
  double* tk_w = reinterpret_cast<double*>(tk_pe + 64);
  constexpr int NT = WAVES * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = blockIdx.y;
  const long lo = (long)blockIdx.x * slab;
  long hi = lo + slab;
  if (hi > n_docs) hi = n_docs;
  const int m = (int)(hi - lo);

  for (int i = tid; i < m; i += NT) sc[i] = 0.0;
  block_sync<WAVES>();

  // Token metadata (posting range inside this slab, idf) is fetched by the lanes IN PARALLEL,
  // 64 tokens at a time, and parked in LDS: walking the tokens one by one would chain three
  // dependent global loads (term id -> term_ptr -> postings) per token, ~1.5 us each.
  const long t0 = q_ptr[qi], t1 = q_ptr[qi + 1];
  for (long tb = t0; tb < t1; tb += 64) {
    const int nt = (int)((t1 - tb) < 64 ? (t1 - tb) : 64);
    if (tid < 64) {
      long ps = 0, pe = 0;
      double w = 0.0;
      if (tid < nt) {
        const int term = q_terms[tb + tid];
        if (term >= 0 && term < n_terms) {  // unknown token: idf 0, contributes +0.0 -> skipped
          const long p0 = term_ptr[term], p1 = term_ptr[term + 1];
          ps = (lo == 0) ? p0 : lower_bound_i32(post_doc, p0, p1, (int)lo);
          pe = (hi >= n_docs) ? p1 : lower_bound_i32(post_doc, ps, p1, (int)hi);
          w = idf[term];
        }
      }
      tk_ps[tid] = ps;
      tk_pe[tid] = pe;
      tk_w[tid] = w;
    }
    block_sync<WAVES>();
    for (int t = 0; t < nt; ++t) {  // query order: this is the accumulation order of rank_bm25
      const long ps = tk_ps[t], pe = tk_pe[t];
      const double w = tk_w[t];
      // rank_bm25: idf * (q_freq * (k1 + 1) / (q_freq + k1 * (1 - b + b * doc_len / avgdl))); the
      // parenthesis depends only on (tf, doc) and was evaluated once at index creation.
      // Four posting chunks are loaded before the first is applied (a list holds a document once,
      // so the scatter has no conflicts and needs no ordering inside one token).
      for (long p = ps + tid; p < pe; p += 4 * NT) {
        int dd[4];
        double ww[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long pp = p + (long)u * NT;
          const bool ok = pp < pe;
          dd[u] = ok ? post_doc[pp] : -1;
          ww[u] = ok ? post_w[pp] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (dd[u] >= 0) sc[dd[u] - lo] += w * ww[u];
      }
      block_sync<WAVES>();
    }
  }

  if (scores_out) {
    for (int i = tid; i < m; i += NT) scores_out[(size_t)qi * n_docs + lo + i] = sc[i];
  }
  if (!part && !fin_ids) return;

  WaveTopK<C64> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  // Short slab and shallow k (the serving shape: <= 1024 docs, k <= 16): k rounds of a wave-wide
  // arg-max directly on the fp64 scores held in registers (16 per lane).  A round = lane-local
  // v_max_f64 over 16 values, a 6-step butterfly, then the LOWEST position holding that value
  // (ties -> lower doc id, as the stable sort of bm25_retriever.py:75) and its removal: ~80
  // instructions, vs two LDS bitonic sorts of 128-bit candidates for the staged selector
  // (measured 65 of the kernel's 98 us per 9 344 queries).  The register selector of topk.hpp
  // was tried here too: with 128-bit candidates its two shuffle networks were slower still.
  bool done = false;
  if (WAVES == 1 && k <= 16 && m <= 1024) {
    const double ninf = -INFINITY;
    double sv[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int i = lane + 64 * v;
      double x = (i < m) ? sc[i] + 0.0 : ninf;  // -0.0 -> +0.0
      sv[v] = (x != x) ? -DBL_MAX : x;          // NaN ranks last among real documents
    }
    int got = 0;
    for (int it = 0; it < k; ++it) {
      double lm = sv[0];
#pragma unroll
      for (int v = 1; v < 16; ++v) lm = fmax(lm, sv[v]);
      double wm = lm;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wm = fmax(wm, __shfl_xor(wm, o));
      if (wm == ninf) break;  // fewer than k documents in the slab
      int pos = 0x7fffffff;
#pragma unroll
      for (int v = 15; v >= 0; --v)
        if (sv[v] == wm) pos = lane + 64 * v;
      int win = pos;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(win, o);
        win = other < win ? other : win;
      }
#pragma unroll
      for (int v = 0; v < 16; ++v)
        if (lane + 64 * v == win) sv[v] = ninf;
      if (lane == 0) tk.buf[it] = C64::make(wm, lo + win);
      got = it + 1;
    }
    wave_lds_fence();
    tk.cnt = got;
    done = true;
  }
  if (!done) {
    for (int base = wave * 64; base < m; base += WAVES * 64) {
      int i = base + lane;
      bool v = i < m;
      C64 c = v ? C64::make(sc[i], lo + i) : C64::pad();
      tk.push_lanes(c, v, lane);
    }
    tk.finalize(lane);
  }
  if (WAVES > 1) block_combine_topk(tk, lists, cap, WAVES, wave, lane, cnts);
  if (wave == 0) {
    if (fin_ids) {
      for (int j = lane; j < k; j += 64) {
        bool v = j < tk.cnt;
        fin_scores[(size_t)qi * k + j] = v ? unord64(tk.buf[j].key) : -DBL_MAX;
        fin_ids[(size_t)qi * k + j] = v ? tk.buf[j].idv : -1ll;
      }
    } else {
      C64* dst = part + ((size_t)blockIdx.x * nq + qi) * k;
      for (int j = lane; j < k; j += 64) dst[j] = (j < tk.cnt) ? tk.buf[j] : C64::pad();
    }
  }
}

__global__ __launch_bounds__(256) void bm25_merge_kernel(const C64* __restrict__ part, int nparts, int nq, int k,
                                                          int cap, double* __restrict__ out_scores,
                                                          long long* __restrict__ out_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C64* lists = reinterpret_cast<C64*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)kBmWaves * cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x;
  WaveTopK<C64> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  const long total = (long)nparts * k;
  for (long base = (long)wave * 64; base < total; base += (long)kBmWaves * 64) {
    long i = base + lane;
    bool v = i < total;
    C64 c = C64::pad();
    if (v) {
      long p = i / k, j = i - p * k;
      c = part[((size_t)p * nq + qi) * k + j];
      v = !c.is_pad();
    }
    tk.push_lanes(c, v, lane);
  }
  tk.finalize(lane);
  block_combine_topk(tk, lists, cap, kBmWaves, wave, lane, cnts);
  if (wave == 0) {
    for (int j = lane; j < k; j += 64) {
      bool v = j < tk.cnt;
      C64 c = v ? tk.buf[j] : C64::pad();
      out_scores[(size_t)qi * k + j] = v ? unord64(c.key) : -DBL_MAX;
      out_ids[(size_t)qi * k + j] = v ? c.idv : -1ll;
    }
  }
}

}  // namespace amdr

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
  DevBuf part, qterms, qptr, sbuf, ibuf, full;
};

namespace {

constexpr int kSlabMax = 4096;  // docs per block: 32 KiB of fp64 scores in LDS

struct BmPlan {
  int slab, nslabs, cap, waves;
  size_t lds, part_bytes;
};

void bm_plan(int64_t n_docs, int nq, int k, BmPlan* p) {
  p->cap = topk_cap(k);
  p->slab = n_docs < kSlabMax ? (int)(n_docs > 0 ? n_docs : 1) : kSlabMax;
  p->nslabs = n_docs > 0 ? (int)((n_docs + p->slab - 1) / p->slab) : 1;
  p->waves = p->slab <= 1024 ? 1 : kBmWaves;
  p->lds = (size_t)p->slab * sizeof(double) + (size_t)p->waves * p->cap * sizeof(C64) + 4 * sizeof(int) +
           64 * (2 * sizeof(long) + sizeof(double));
  p->part_bytes = (size_t)p->nslabs * nq * k * sizeof(C64);
}

int bm_run(amdr_bm25* h, const int* q_terms_dev, const long long* q_ptr_dev, int nq, int k, double* scores_dev,
           int64_t* ids_dev, double* full_dev, hipStream_t st) {
  BmPlan p;
  bm_plan(h->n_docs, nq, k, &p);
  C64* part = nullptr;
  if (scores_dev) {
    int rc = h->part.ensure(p.part_bytes);
    if (rc) return rc;
    part = h->part.as<C64>();
  }
  const bool direct = scores_dev && p.nslabs == 1;  // single slab: the block's list is the answer
  double* fs = direct ? scores_dev : nullptr;
  long long* fi = direct ? (long long*)ids_dev : nullptr;
  if (direct) part = nullptr;
  if (p.waves == 1)
    hipLaunchKernelGGL(bm25_score_topk_kernel<1>, dim3(p.nslabs, nq), dim3(64), p.lds, st, h->term_ptr, h->post_doc,
                       h->post_w, h->idf, (long)h->n_terms, (long)h->n_docs, q_terms_dev, q_ptr_dev, nq, k, p.cap,
                       p.slab, full_dev, part, fs, fi);
  else
    hipLaunchKernelGGL(bm25_score_topk_kernel<kBmWaves>, dim3(p.nslabs, nq), dim3(256), p.lds, st, h->term_ptr,
                       h->post_doc, h->post_w, h->idf, (long)h->n_terms, (long)h->n_docs, q_terms_dev, q_ptr_dev, nq,
                       k, p.cap, p.slab, full_dev, part, fs, fi);
  AMDR_HIP(hipGetLastError());
  if (scores_dev && !direct) {
    size_t lds = (size_t)kBmWaves * p.cap * sizeof(C64) + kBmWaves * sizeof(int);
    hipLaunchKernelGGL(bm25_merge_kernel, dim3(nq), dim3(256), lds, st, part, p.nslabs, nq, k, p.cap, scores_dev,
                       (long long*)ids_dev);
    AMDR_HIP(hipGetLastError());
  }
  return AMDR_OK;
}

template <class T>
int upload(T** dst, const T* src, size_t count) {
  *dst = nullptr;
  size_t bytes = (count ? count : 1) * sizeof(T);
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
  int rc = h->part.ensure(p.part_bytes);
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
  return bm_run(h, q_terms_dev, (const long long*)q_ptr_dev, nq, k, scores_dev, ids_dev, nullptr, (hipStream_t)stream);
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
  rc = bm_run(h, h->qterms.as<int>(), h->qptr.as<long long>(), nq, k, h->sbuf.as<double>(), h->ibuf.as<int64_t>(),
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
  rc = bm_run(h, h->qterms.as<int>(), h->qptr.as<long long>(), nq, 1, nullptr, nullptr, h->full.as<double>(),
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
  h->part.release();
  h->qterms.release();
  h->qptr.release();
  h->sbuf.release();
  h->ibuf.release();
  h->full.release();
  delete h;
  return AMDR_OK;
}

}  // extern "C"
