// ColBERT channel: exhaustive late-interaction MaxSim for gfx950.
//
// Replaces `Searcher.search(query, k)` (legalrag/retrieval/colbert_retriever.py:152):
//   score(q, doc) = sum_{i < q_len} max_{j < len(doc)} <q_i, d_j>,  dim = 128, fp32.
// A wave scores (query, document) pairs tile by tile: the 32 query tokens x 32 document tokens
// similarity tile is a genuine small matrix product, so it runs on the matrix cores with the
// fp32-input v_mfma_f32_16x16x4_f32 (exact fp32 products and sums, the peak rate of the vector
// ALU, no cross-lane reduction, no LDS broadcast traffic), four 16x16 accumulator blocks:
//   A (16 doc tokens x 4)   : lane (i16 = l&15, kq = l>>4) holds D[tok0 + 16 bi + i16][16-B slots 4t + kq]
//   B (4 x 16 query tokens) : lane (i16, kq)              holds Q[16 bj + i16][16-B slots 4t + kq]
// (t = 0..7: the k index is permuted, identically for both operands — a dot product does not
// care.)  C[doc token][query token] comes back with the query token on the lane (l & 15) and
// 4 doc tokens per accumulator block in the lane's registers, so max-over-tokens is in-register
// plus two lane exchanges, and the final sum over query tokens is one DPP reduction per document.
// (First built on v_mfma_f32_32x32x2_f32; the 16x16 form holds a higher clock, see dense_mfma.hip.)
// Algorithmic bytes per (query, shard): sum_docs len*128*4; flops 2*32*128*sum len.
#include "common.hpp"
#include "topk.hpp"

#include <cfloat>
#include <mutex>
#include <new>

namespace amdr {


constexpr int kMsWaves = 4;
constexpr int kDim = AMDR_MAXSIM_DIM;  // 128

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ms_dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float ms_wave_sum(float v) {  // total in lane 63
  v = ms_dpp_add<0x111, 0xf>(v);
  v = ms_dpp_add<0x112, 0xf>(v);
  v = ms_dpp_add<0x114, 0xf>(v);
  v = ms_dpp_add<0x118, 0xf>(v);
  v = ms_dpp_add<0x142, 0xa>(v);
  v = ms_dpp_add<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float ms4f __attribute__((ext_vector_type(4)));

// One 32x32 tile: 8 slots x 4 components x (2 x 2 accumulator blocks) = 128 MFMAs of 32 cycles.
// a[bi][t] / q[bj][t]: the lane's 16-B slot 4t + kq of document-token row 16 bi + i16 / query-token
// row 16 bj + i16.  Returns, per query-token block bj, the maximum over this lane's 8 document
// tokens (rows 16 bi + 4 kq + r), rows >= remain masked out.
// NBI = 1: the tile holds at most 16 document tokens (the tail of a document): the second row block
// would be masked out entirely, so its 64 MFMAs (and the caller's 8 fragment reads) are skipped —
// same results, and documents are short (Civil-Code articles: 77 tokens on average, 17 % of the
// 32-token tile slots were padding).
template <int NBI>
__device__ __forceinline__ void ms_tile(const ms4f (&a)[2][8], const ms4f (&q)[2][8], int kq, int remain,
                                        float (&best)[2]) {
  f32x4 acc[NBI][2];
#pragma unroll
  for (int bi = 0; bi < NBI; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int bi = 0; bi < NBI; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bi][t][e], q[bj][t][e], acc[bi][bj], 0, 0, 0);
#pragma unroll
  for (int bi = 0; bi < NBI; ++bi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool off = remain < 32 && (16 * bi + 4 * kq + r) >= remain;
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) best[bj] = fmaxf(best[bj], off ? -FLT_MAX : acc[bi][bj][r]);
    }
}

// Document score from the per-lane maxima: max over the four kq groups, then sum over the
// q_len query tokens (lane group kq = 0 holds token 16 bj + i16 in best[bj]).
__device__ __forceinline__ float ms_finish(const float (&best)[2], int i16, int kq, int q_len) {
  float contrib = 0.f;
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) {
    float b = best[bj];
    b = fmaxf(b, __uint_as_float(lane_xor<16>(__float_as_uint(b))));
    b = fmaxf(b, __uint_as_float(lane_xor<32>(__float_as_uint(b))));
    if (kq == 0 && 16 * bj + i16 < q_len) contrib += b;
  }
  return ms_wave_sum(contrib);
}

// The lane's fragment of a 32-row x 128-float operand in global memory: rows row0 + 16 b + i16
// (clamped to row_max), 16-B slots 4t + kq.
__device__ __forceinline__ void ms_load_frag(const float* __restrict__ base, long row0, long row_max, int i16, int kq,
                                             bool zero, ms4f (&f)[2][8]) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    long row = row0 + 16 * b + i16;
    const bool out = zero && row > row_max;
    if (row > row_max) row = row_max;
    const float* p = base + (size_t)row * kDim + kq * 4;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const ms4f v = *reinterpret_cast<const ms4f*>(p + 16 * t);
      f[b][t] = out ? ms4f{0.f, 0.f, 0.f, 0.f} : v;
    }
  }
}

// grid: (x = ceil(n_docs / 4), y = queries); one wave per document (1-7 queries: latency form).
__global__ __launch_bounds__(256) void maxsim_scores_kernel(const float* __restrict__ D,
                                                             const long long* __restrict__ doc_ptr, long n_docs,
                                                             const float* __restrict__ Q, int q_len,
                                                             float* __restrict__ scores /*[nq, n_docs]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long doc = (long)blockIdx.x * kMsWaves + wave;
  if (doc >= n_docs) return;  // whole wave exits together
  const int qi = blockIdx.y;
  const int i16 = lane & 15, kq = lane >> 4;

  ms4f qf[2][8];  // query tokens past q_len are zero rows
  ms_load_frag(Q + (size_t)qi * q_len * kDim, 0, q_len - 1, i16, kq, true, qf);

  const long t_lo = doc_ptr[doc], t_hi = doc_ptr[doc + 1];
  const int len = (int)(t_hi - t_lo);
  float best[2] = {-FLT_MAX, -FLT_MAX};
  for (int tok0 = 0; tok0 < len; tok0 += 32) {
    ms4f af[2][8];  // rows past the document end are clamped here and masked in ms_tile
    ms_load_frag(D, t_lo + tok0, t_hi - 1, i16, kq, false, af);
    if (len - tok0 <= 16)
      ms_tile<1>(af, qf, kq, len - tok0, best);
    else
      ms_tile<2>(af, qf, kq, len - tok0, best);
  }
  const float total = ms_finish(best, i16, kq, q_len);
  if (lane == 0) scores[(size_t)qi * n_docs + doc] = total;
}


// Blocked form for query batches: a block = 8 waves = 8 queries, and walks kMsDocs documents.
// The one-wave-per-(query, document) kernel above re-reads every document's tokens for every
// query (PMC, UCC-en step of 1 168 queries: 23-46 GB of L2-miss reads against 68 MB of
// algorithmic bytes); here a 32-token document tile is fetched ONCE per block with coalesced
// 16-B/lane loads into a double-buffered, XOR-swizzled 16-KiB LDS tile and feeds all eight
// queries' MFMAs (query fragments live in registers for the whole block); the next tile —
// across document boundaries — is in flight while the current one is multiplied.  Blocks that
// share a document group have consecutive ids (query group = fast grid index), so they run
// together and the group stays in every XCD's L2.  Same MFMA operands and k order as above:
// bit-identical scores.
constexpr int kMsQ = 8;     // queries (waves) per block
constexpr int kMsDocs = 8;  // documents per block

// LDS tile: token row j (0..31) at byte j*512, its 16-B slot s (0..31) at s ^ (j & 15).  A
// ds_read_b128 lane group holds 16 distinct rows, eight reading slot 4t + kq and eight
// 4t + (kq ^ 1): the XOR maps those two sets onto disjoint bank quads (conflict-free); a
// staging write of one row (32 consecutive threads) covers the row's 512 B.
__device__ __forceinline__ int ms_tile_off(int row, int slot) { return row * 512 + ((slot ^ (row & 15)) << 4); }

__global__ __launch_bounds__(kMsQ * 64) void maxsim_scores_blocked_kernel(const float* __restrict__ D,
                                                                           const long long* __restrict__ doc_ptr,
                                                                           long n_docs, long n_tokens,
                                                                           const float* __restrict__ Q, int nq, int q_len,
                                                                           float* __restrict__ scores /*[nq, n_docs]*/) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[2][32 * 512];
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int qi = blockIdx.x * kMsQ + wave;
  const bool live = qi < nq;
  const long d0 = (long)blockIdx.y * kMsDocs;
  long d1 = d0 + kMsDocs;
  if (d1 > n_docs) d1 = n_docs;

  ms4f qf[2][8];
  if (live) {
    ms_load_frag(Q + (size_t)qi * q_len * kDim, 0, q_len - 1, i16, kq, true, qf);
  } else {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < 8; ++t) qf[b][t] = ms4f{0.f, 0.f, 0.f, 0.f};
  }

  // loader role: two 16-B pieces per thread and tile (elements tid and tid + 512 of 1024)
  const int lrow0 = tid >> 5, lslot = tid & 31;  // rows lrow0 and lrow0 + 16
  long doc = d0;
  long t_lo = doc_ptr[doc];
  int len = (int)(doc_ptr[doc + 1] - t_lo);
  int tok0 = 0;
  v4f g[2];
#define AMDR_MS_LOAD(TLO, LEN, TOK0)                                                                   \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                      \
    int j_ = (TOK0) + lrow0 + 16 * u;                                                                  \
    if (j_ >= (LEN)) j_ = (LEN)-1; /* rows past the document end are masked after the MFMAs */         \
    g[u] = *reinterpret_cast<const v4f*>(D + (size_t)((TLO) + j_) * kDim + lslot * 4);                 \
  }
#define AMDR_MS_STAGE(BUF)                                                                             \
  _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                        \
      *reinterpret_cast<v4f*>(tile[BUF] + ms_tile_off(lrow0 + 16 * u, lslot)) = g[u];
  AMDR_MS_LOAD(t_lo, len, tok0)
  AMDR_MS_STAGE(0)
  __syncthreads();
  int buf = 0;
  float best[2] = {-FLT_MAX, -FLT_MAX};
  (void)n_tokens;
  while (true) {
    // coordinates of the next tile (wave-uniform)
    long ndoc = doc;
    int ntok = tok0 + 32;
    long nt_lo = t_lo;
    int nlen = len;
    if (ntok >= len) {
      ndoc = doc + 1;
      ntok = 0;
      if (ndoc < d1) {
        nt_lo = doc_ptr[ndoc];
        nlen = (int)(doc_ptr[ndoc + 1] - nt_lo);
      }
    }
    const bool has_next = ndoc < d1;
    if (has_next) { AMDR_MS_LOAD(nt_lo, nlen, ntok) }

    ms4f af[2][8];
    if (len - tok0 <= 16) {  // wave-uniform: the tail of a document fits one 16-token row block
#pragma unroll
      for (int t = 0; t < 8; ++t)
        af[0][t] = *reinterpret_cast<const ms4f*>(tile[buf] + ms_tile_off(i16, 4 * t + kq));
      ms_tile<1>(af, qf, kq, len - tok0, best);
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 8; ++t)
          af[b][t] = *reinterpret_cast<const ms4f*>(tile[buf] + ms_tile_off(16 * b + i16, 4 * t + kq));
      ms_tile<2>(af, qf, kq, len - tok0, best);
    }
    if (ntok == 0) {  // last tile of this document
      const float total = ms_finish(best, i16, kq, q_len);
      if (live && lane == 0) scores[(size_t)qi * n_docs + doc] = total;
      best[0] = best[1] = -FLT_MAX;
    }
    if (!has_next) break;
    AMDR_MS_STAGE(buf ^ 1)
    __syncthreads();
    buf ^= 1;
    doc = ndoc;
    tok0 = ntok;
    t_lo = nt_lo;
    len = nlen;
  }
#undef AMDR_MS_LOAD
#undef AMDR_MS_STAGE
}

// Per-query top-k over a dense fp32 score row (one block per query).
__global__ __launch_bounds__(256) void rowscores_topk_kernel(const float* __restrict__ scores, long n, int k, int cap,
                                                              float* __restrict__ out_scores,
                                                              long long* __restrict__ out_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)kMsWaves * cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x;
  const float* row = scores + (size_t)qi * n;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  for (long base = (long)wave * 64; base < n; base += (long)kMsWaves * 64) {
    long i = base + lane;
    bool v = i < n;
    C32 c = v ? C32::make(row[i], (u32)i) : C32::pad();
    tk.push_lanes(c, v, lane);
  }
  tk.finalize(lane);
  block_combine_topk(tk, lists, cap, kMsWaves, wave, lane, cnts);
  if (wave == 0) {
    for (int j = lane; j < k; j += 64) {
      bool v = j < tk.cnt;
      C32 c = v ? tk.buf[j] : C32::pad();
      out_scores[(size_t)qi * k + j] = v ? c.score() : -FLT_MAX;
      out_ids[(size_t)qi * k + j] = v ? c.id() : -1ll;
    }
  }
}

}  // namespace amdr

using namespace amdr;

struct amdr_maxsim {
  int device = 0;
  int64_t n_docs = 0, n_tokens = 0;
  float* D = nullptr;
  long long* doc_ptr = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  DevBuf full[2], qbuf, sbuf, ibuf;  // full[0]: "_device" calls, full[1]: host-pointer calls (see dense.hip)
};

namespace {

int ms_run(amdr_maxsim* h, const float* Q_dev, int nq, int q_len, int k, float* full_dev, float* scores_dev,
           int64_t* ids_dev, hipStream_t st) {
  if (nq >= kMsQ && ceil_div(h->n_docs, kMsDocs) <= 65535) {  // batches: document tiles shared by 8 queries through LDS
    dim3 grid(ceil_div(nq, kMsQ), ceil_div(h->n_docs, kMsDocs));
    hipLaunchKernelGGL(maxsim_scores_blocked_kernel, grid, dim3(kMsQ * 64), 0, st, h->D, h->doc_ptr, (long)h->n_docs,
                       (long)h->n_tokens, Q_dev, nq, q_len, full_dev);
  } else {
    dim3 grid(ceil_div(h->n_docs, kMsWaves), nq);
    hipLaunchKernelGGL(maxsim_scores_kernel, grid, dim3(256), 0, st, h->D, h->doc_ptr, (long)h->n_docs, Q_dev, q_len,
                       full_dev);
  }
  AMDR_HIP(hipGetLastError());
  if (scores_dev) {
    int cap = topk_cap(k);
    size_t lds = (size_t)kMsWaves * cap * sizeof(C32) + kMsWaves * sizeof(int);
    hipLaunchKernelGGL(rowscores_topk_kernel, dim3(nq), dim3(256), lds, st, full_dev, (long)h->n_docs, k, cap,
                       scores_dev, (long long*)ids_dev);
    AMDR_HIP(hipGetLastError());
  }
  return AMDR_OK;
}

int ms_check(const amdr_maxsim* h, const void* Q, int nq, int q_len, int k) {
  AMDR_REQUIRE(h != nullptr, "maxsim: null handle");
  AMDR_REQUIRE(nq >= 0, "maxsim: nq=%d", nq);
  AMDR_REQUIRE(q_len >= 1 && q_len <= AMDR_MAXSIM_QLEN, "maxsim: q_len=%d outside [1,%d]", q_len, AMDR_MAXSIM_QLEN);
  AMDR_REQUIRE(k >= 1 && k <= AMDR_MAX_K, "maxsim: k=%d outside [1,%d]", k, AMDR_MAX_K);
  AMDR_REQUIRE(nq == 0 || Q, "maxsim: null Q");
  return AMDR_OK;
}

}  // namespace

extern "C" {

int amdr_maxsim_create(const float* D_host, const int64_t* doc_ptr, int64_t n_docs, int32_t dim, int32_t device,
                       amdr_maxsim_t** out) {
  AMDR_REQUIRE(out != nullptr, "maxsim_create: out is null");
  *out = nullptr;
  AMDR_REQUIRE(dim == AMDR_MAXSIM_DIM, "maxsim_create: dim=%d, kernel is built for %d", dim, AMDR_MAXSIM_DIM);
  AMDR_REQUIRE(doc_ptr && n_docs >= 1 && n_docs < (1ll << 32), "maxsim_create: bad doc_ptr / n_docs");
  AMDR_REQUIRE(doc_ptr[0] == 0, "maxsim_create: doc_ptr[0] != 0");
  for (int64_t i = 0; i < n_docs; ++i)
    AMDR_REQUIRE(doc_ptr[i + 1] > doc_ptr[i], "maxsim_create: document %lld has no tokens", (long long)i);
  const int64_t nt = doc_ptr[n_docs];
  AMDR_REQUIRE(D_host != nullptr, "maxsim_create: D is null");
  int rc = check_device(device);
  if (rc) return rc;
  amdr_maxsim* h = new (std::nothrow) amdr_maxsim();
  if (!h) return fail(AMDR_ENOMEM, "maxsim_create: host alloc");
  h->device = device;
  h->n_docs = n_docs;
  h->n_tokens = nt;
  hipError_t e = hipMalloc((void**)&h->D, (size_t)nt * dim * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(h->D, D_host, (size_t)nt * dim * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&h->doc_ptr, (size_t)(n_docs + 1) * sizeof(long long));
  if (e == hipSuccess)
    e = hipMemcpy(h->doc_ptr, doc_ptr, (size_t)(n_docs + 1) * sizeof(long long), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    amdr_maxsim_destroy(h);
    return fail(e == hipErrorOutOfMemory ? AMDR_ENOMEM : AMDR_EHIP, "maxsim_create: %s", hipGetErrorString(e));
  }
  *out = h;
  return AMDR_OK;
}

int amdr_maxsim_ndocs(const amdr_maxsim_t* h, int64_t* n) {
  AMDR_REQUIRE(h && n, "maxsim_ndocs: null");
  *n = h->n_docs;
  return AMDR_OK;
}

int amdr_maxsim_reserve(amdr_maxsim_t* h, int32_t nq_max, int32_t k_max) {
  AMDR_REQUIRE(h != nullptr, "maxsim_reserve: null handle");
  AMDR_REQUIRE(nq_max >= 1 && k_max >= 1 && k_max <= AMDR_MAX_K, "maxsim_reserve: bad sizes");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  int rc = h->full[0].ensure((size_t)nq_max * h->n_docs * sizeof(float));
  if (!rc) rc = h->qbuf.ensure((size_t)nq_max * AMDR_MAXSIM_QLEN * kDim * sizeof(float));
  if (!rc) rc = h->sbuf.ensure((size_t)nq_max * k_max * sizeof(float));
  if (!rc) rc = h->ibuf.ensure((size_t)nq_max * k_max * sizeof(int64_t));
  return rc;
}

int amdr_maxsim_search_device(amdr_maxsim_t* h, const float* Q_dev, int32_t nq, int32_t q_len, int32_t k,
                              float* scores_dev, int64_t* ids_dev, void* stream) {
  int rc = ms_check(h, Q_dev, nq, q_len, k);
  if (rc) return rc;
  AMDR_REQUIRE(nq == 0 || (scores_dev && ids_dev), "maxsim: null output");
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  if ((rc = h->full[0].ensure((size_t)nq * h->n_docs * sizeof(float)))) return rc;
  return ms_run(h, Q_dev, nq, q_len, k, h->full[0].as<float>(), scores_dev, ids_dev, (hipStream_t)stream);
}

int amdr_maxsim_search(amdr_maxsim_t* h, const float* Q_host, int32_t nq, int32_t q_len, int32_t k,
                       float* scores_host, int64_t* ids_host) {
  int rc = ms_check(h, Q_host, nq, q_len, k);
  if (rc) return rc;
  AMDR_REQUIRE(nq == 0 || (scores_host && ids_host), "maxsim: null output");
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  size_t qbytes = (size_t)nq * q_len * kDim * sizeof(float);
  if ((rc = h->qbuf.ensure(qbytes))) return rc;
  if ((rc = h->full[1].ensure((size_t)nq * h->n_docs * sizeof(float)))) return rc;
  if ((rc = h->sbuf.ensure((size_t)nq * k * sizeof(float)))) return rc;
  if ((rc = h->ibuf.ensure((size_t)nq * k * sizeof(int64_t)))) return rc;
  AMDR_HIP(hipMemcpyAsync(h->qbuf.p, Q_host, qbytes, hipMemcpyHostToDevice, h->stream));
  rc = ms_run(h, h->qbuf.as<float>(), nq, q_len, k, h->full[1].as<float>(), h->sbuf.as<float>(), h->ibuf.as<int64_t>(),
              h->stream);
  if (rc) return rc;
  AMDR_HIP(hipMemcpyAsync(scores_host, h->sbuf.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipMemcpyAsync(ids_host, h->ibuf.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipStreamSynchronize(h->stream));
  return AMDR_OK;
}

int amdr_maxsim_scores(amdr_maxsim_t* h, const float* Q_host, int32_t nq, int32_t q_len, float* scores_host) {
  int rc = ms_check(h, Q_host, nq, q_len, 1);
  if (rc) return rc;
  AMDR_REQUIRE(nq == 0 || scores_host, "maxsim_scores: null output");
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  size_t qbytes = (size_t)nq * q_len * kDim * sizeof(float);
  if ((rc = h->qbuf.ensure(qbytes))) return rc;
  if ((rc = h->full[1].ensure((size_t)nq * h->n_docs * sizeof(float)))) return rc;
  AMDR_HIP(hipMemcpyAsync(h->qbuf.p, Q_host, qbytes, hipMemcpyHostToDevice, h->stream));
  rc = ms_run(h, h->qbuf.as<float>(), nq, q_len, 1, h->full[1].as<float>(), nullptr, nullptr, h->stream);
  if (rc) return rc;
  AMDR_HIP(hipMemcpyAsync(scores_host, h->full[1].p, (size_t)nq * h->n_docs * sizeof(float), hipMemcpyDeviceToHost,
                          h->stream));
  AMDR_HIP(hipStreamSynchronize(h->stream));
  return AMDR_OK;
}

int amdr_maxsim_destroy(amdr_maxsim_t* h) {
  if (!h) return AMDR_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamDestroy(h->stream);
  }
  if (h->D) (void)hipFree(h->D);
  if (h->doc_ptr) (void)hipFree(h->doc_ptr);
  h->full[0].release();
  h->full[1].release();
  h->qbuf.release();
  h->sbuf.release();
  h->ibuf.release();
  delete h;
  return AMDR_OK;
}

}  // extern "C"
