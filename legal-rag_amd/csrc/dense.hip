// Dense channel: exact inner-product scan + fused top-k for gfx950.
//
// Replaces faiss `index.search(q_vec, k)` (legalrag/retrieval/dense_retriever.py:42).
// The chunk matrix X[n, d] (fp32, row-major) is streamed from HBM exactly once
// per pass of NQ queries: a wave owns a row at a time, lane l holds the float4
// pieces at columns 4l + 256c, the NQ query vectors live in registers, and the
// per-row partial sums are folded with DPP row operations (no LDS round trip).
// Scores never go to memory: each wave keeps its own top-k staging buffer in
// LDS (topk.hpp), one list per block is written out and a second small kernel
// merges the per-block lists.  HBM-bound by construction: algorithmic bytes
// = n*d*4 per pass, flops = 2*n*d*NQ.
#include "common.hpp"
#include "dense_dot.hpp"
#include "topk.hpp"

#include <cfloat>
#include <mutex>
#include <new>
#include <vector>

namespace amdr {

constexpr int kWaves = 4;  // 256-thread blocks

// grid: (x = row slabs, y = query groups of NQ).  LDS: kWaves*NQ*cap C32 + ints.
// NT: the chunk matrix is read with the non-temporal cache policy.  A matrix that does not fit the
// 256 MiB Infinity Cache is read once per scan and gains nothing from being kept: measured on
// 10 M x 768 (30.7 GB), one query per scan 4.85-4.96 -> 4.44-4.58 ms (6.3 -> 6.7-6.9 TB/s), four queries
// 4.98 -> 4.64 ms.  Smaller matrices keep the default policy (a repeated scan is then served on-die).
template <int NQ, int CH, int U, bool NT>
__global__ __launch_bounds__(256) void dense_scan_topk_kernel(const float* __restrict__ X, long n, int d,
                                                               const float* __restrict__ Q, int nq_total, int k,
                                                               int cap, long rows_per_block,
                                                               C32* __restrict__ part /*[gridDim.x][nq_total][k]*/,
                                                               float* __restrict__ fin_scores /* single slab */,
                                                               long long* __restrict__ fin_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)kWaves * NQ * cap);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int qbase = blockIdx.y * NQ;

  float4 q[NQ][CH];
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int col = c * 256 + lane * 4;
      bool ok = (qbase + b < nq_total) && (col < d);
      q[b][c] = ok ? *reinterpret_cast<const float4*>(Q + (size_t)(qbase + b) * d + col) : make_float4(0, 0, 0, 0);
    }
  }

  WaveTopK<C32> tk[NQ];
#pragma unroll
  for (int b = 0; b < NQ; ++b) tk[b].init(lists + ((size_t)b * kWaves + wave) * cap, cap, k);

  const long row_lo = (long)blockIdx.x * rows_per_block;
  long row_hi = row_lo + rows_per_block;
  if (row_hi > n) row_hi = n;

  for (long r0 = row_lo + (long)wave * U; r0 < row_hi; r0 += (long)kWaves * U) {
    float4 x[U][CH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long r = r0 + u;
      if (r >= row_hi) r = row_hi - 1;  // clamp: keeps the loads unconditional
      const float* xr = X + (size_t)r * d;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        int col = c * 256 + lane * 4;
        if (NT) {
          typedef float nt4 __attribute__((ext_vector_type(4)));
          const nt4 t_ = (col < d) ? __builtin_nontemporal_load(reinterpret_cast<const nt4*>(xr + col)) : nt4{0, 0, 0, 0};
          x[u][c] = make_float4(t_.x, t_.y, t_.z, t_.w);
        } else {
          x[u][c] = (col < d) ? *reinterpret_cast<const float4*>(xr + col) : make_float4(0, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + u;
      const bool rv = r < row_hi;
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) acc = dot4(x[u][c], q[b][c], acc);
        acc = wave_sum_to_lane63(acc);
        float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63));
        if (rv) tk[b].push_uniform(C32::make(s, (u32)r), lane);
      }
    }
  }

#pragma unroll
  for (int b = 0; b < NQ; ++b) tk[b].finalize(lane);
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    block_combine_topk(tk[b], lists + (size_t)b * kWaves * cap, cap, kWaves, wave, lane, cnts);
    if (wave == 0 && qbase + b < nq_total) {
      if (fin_ids) {  // single slab: this list is the final answer, no merge launch follows
        for (int j = lane; j < k; j += 64) {
          const bool v = j < tk[b].cnt;
          const C32 c = v ? tk[b].buf[j] : C32::pad();
          fin_scores[(size_t)(qbase + b) * k + j] = v ? c.score() : -FLT_MAX;
          fin_ids[(size_t)(qbase + b) * k + j] = v ? c.id() : -1ll;
        }
      } else {
        C32* dst = part + ((size_t)blockIdx.x * nq_total + (qbase + b)) * k;
        for (int j = lane; j < k; j += 64) dst[j] = (j < tk[b].cnt) ? tk[b].buf[j] : C32::pad();
      }
    }
    __syncthreads();
  }
}

// Score an explicit candidate list: out[q][j] = <Q[q], X[rows[q][j]]> (rows < 0 -> -FLT_MAX).
// One wave per (query, candidate): the row is gathered with three coalesced 1-KiB loads.
// Stands where GraphRetriever re-embeds its candidates per query and takes cosines
// (legalrag/retrieval/graph_retriever.py:177-191): the chunk embeddings are already in HBM.
__global__ __launch_bounds__(256) void dense_score_rows_kernel(const float* __restrict__ X, long n, int d,
                                                                const float* __restrict__ Q, int nq,
                                                                const long long* __restrict__ rows, int m,
                                                                float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * kWaves + wave;
  if (idx >= (long)nq * m) return;
  const int qi = (int)(idx / m);
  const long long r = rows[idx];
  if (r < 0 || r >= n) {
    if (lane == 0) out[idx] = -FLT_MAX;
    return;
  }
  const float acc = dense_row_dot(X + (size_t)r * d, Q + (size_t)qi * d, d, lane);
  if (lane == 63) out[idx] = acc;
}

// Short corpus, 1-4 queries (the single-query serving call at UCC-en / Civil-Code size): one
// wave per (query, row) writes S[q][row]; the slab top-k of dense_mfma.hip ranks it.  A row-slab
// scan with per-wave top-k state is latency-bound here (591 rows = 40 waves walking 15 rows each,
// then a merge launch: 15.6 + 8.1 us); 591 independent waves finish in one memory round trip.
__global__ __launch_bounds__(256) void dense_all_scores_kernel(const float* __restrict__ X, long n, int d,
                                                                const float* __restrict__ Q, int nq, long ldS,
                                                                float* __restrict__ S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * kWaves + wave;
  if (idx >= (long)nq * n) return;
  const int qi = (int)(idx / n);
  const long r = idx - (long)qi * n;
  const float acc = dense_row_dot(X + (size_t)r * d, Q + (size_t)qi * d, d, lane);
  if (lane == 63) S[(size_t)qi * ldS + r] = acc;
}

// One block per query: stream the per-block lists, keep the best k, decode.
__global__ __launch_bounds__(256) void dense_merge_kernel(const C32* __restrict__ part, int nparts, int nq, int k,
                                                           int cap, float* __restrict__ out_scores,
                                                           long long* __restrict__ out_ids,
                                                           const int* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (gate != nullptr && *gate == 0) return;  // gated launch: see run_search_two_level
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)kWaves * cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  const long total = (long)nparts * k;
  for (long base = (long)wave * 64; base < total; base += (long)kWaves * 64) {
    long i = base + lane;
    bool v = i < total;
    C32 c = C32::pad();
    if (v) {
      long p = i / k, j = i - p * k;
      c = part[((size_t)p * nq + qi) * k + j];
      v = !c.is_pad();
    }
    tk.push_lanes(c, v, lane);
  }
  tk.finalize(lane);
  block_combine_topk(tk, lists, cap, kWaves, wave, lane, cnts);
  if (wave == 0) {
    for (int j = lane; j < k; j += 64) {
      bool v = j < tk.cnt;
      C32 c = v ? tk.buf[j] : C32::pad();
      out_scores[(size_t)qi * k + j] = v ? c.score() : -FLT_MAX;
      out_ids[(size_t)qi * k + j] = v ? c.id() : -1ll;
    }
  }
}

// Generic [n_parts, nq, k_in] (score, global id) merge used after the RCCL
// all-gather of per-shard results.  T = float or double.
template <class T>
__global__ __launch_bounds__(256) void merge_parts_kernel(const T* __restrict__ scores,
                                                           const long long* __restrict__ ids, int nparts, int nq,
                                                           int k_in, int k_out, int cap, T* __restrict__ out_scores,
                                                           long long* __restrict__ out_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C64* lists = reinterpret_cast<C64*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)kWaves * cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x;
  WaveTopK<C64> tk;
  tk.init(lists + (size_t)wave * cap, cap, k_out);
  const long total = (long)nparts * k_in;
  for (long base = (long)wave * 64; base < total; base += (long)kWaves * 64) {
    long i = base + lane;
    bool v = i < total;
    C64 c = C64::pad();
    if (v) {
      long p = i / k_in, j = i - p * k_in;
      size_t off = ((size_t)p * nq + qi) * k_in + j;
      long long id = ids[off];
      v = id >= 0;
      if (v) c = sizeof(T) == 8 ? C64::make((double)scores[off], id) : C64::make32((float)scores[off], id);
    }
    tk.push_lanes(c, v, lane);
  }
  tk.finalize(lane);
  block_combine_topk(tk, lists, cap, kWaves, wave, lane, cnts);
  if (wave == 0) {
    for (int j = lane; j < k_out; j += 64) {
      bool v = j < tk.cnt;
      C64 c = v ? tk.buf[j] : C64::pad();
      T s;
      if (sizeof(T) == 8)
        s = v ? (T)unord64(c.key) : (T)(-DBL_MAX);
      else
        s = v ? (T)unord32((u32)c.key) : (T)(-FLT_MAX);
      out_scores[(size_t)qi * k_out + j] = s;
      out_ids[(size_t)qi * k_out + j] = v ? c.idv : -1ll;
    }
  }
}

template <class T>
int launch_merge_parts(const T* scores, const int64_t* ids, int nparts, int nq, int k_in, int k_out, T* out_scores,
                       int64_t* out_ids, hipStream_t st) {
  int cap = topk_cap(k_out);
  size_t lds = (size_t)kWaves * cap * sizeof(C64) + kWaves * sizeof(int);
  hipLaunchKernelGGL((merge_parts_kernel<T>), dim3(nq), dim3(256), lds, st, scores, (const long long*)ids, nparts, nq,
                     k_in, k_out, cap, out_scores, (long long*)out_ids);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}
template int launch_merge_parts<float>(const float*, const int64_t*, int, int, int, int, float*, int64_t*, hipStream_t);
template int launch_merge_parts<double>(const double*, const int64_t*, int, int, int, int, double*, int64_t*,
                                        hipStream_t);

}  // namespace amdr

using namespace amdr;

struct amdr_dense {
  int device = 0;
  int64_t n = 0;
  int64_t cap_rows = 0;
  int d = 0;
  float* X = nullptr;
  bool owns = true;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // Two workspaces: [0] for the "_device" entry points (kernels enqueued on the CALLER's stream,
  // the call returns before they run), [1] for the host-pointer entry points (own stream,
  // synchronised before the mutex is released).  A service thread in amdr_dense_search can
  // therefore never scribble over the score matrix of a search_batch still in flight on another
  // stream.  "_device" calls on ONE handle from SEVERAL streams remain the caller's to order.
  DevBuf part[2], smat[2], aux[2], qbuf, sbuf, ibuf;  // aux: candidate tiles of the two-level top-k
  // matrix statistics for the fp16 first pass of large scans (dense_hi.hip): kept up to date by create / add
  DevBuf stats;
  float x_scale = 1.f;       // power of two: |x| * x_scale < 1 for every component
  float row_norm_max = 0.f;  // largest row L2 norm
  bool hi_ok = false;        // d supported and both statistics finite
  int64_t hi_queries = 0;    // queries that went through the fp16 first pass (amdr_dense_hi_counters)
  // adaptive width of the candidate cut: level l re-scores k + max(k, kHiExtra[l]) + 1 tiles per query; a handle whose
  // queries the rounding bound keeps failing to resolve moves up a level, and at the top level gives the pass up
  int hi_level = 0;
  bool hi_off = false;
  int64_t hi_passes = 0;        // passes (<= 64 queries each) through the fp16 first pass
  int64_t lvl_p0 = 0;           // hi_passes / flagged-pass counter when the current level was entered
  unsigned int lvl_f0 = 0;
  unsigned int* hi_host = nullptr;  // pinned: the device's (unresolved queries, flagged passes, passes), copied back after every search
  hipEvent_t hi_ev = nullptr;       // recorded behind that copy: hi_adapt reads hi_host only once it has completed
  bool hi_copy_pending = false;
  unsigned int hi_seen[3] = {0u, 0u, 0u};  // the last completed copy
  // two-pass long-batch form on a short corpus (dense_small_hi.hip + fuse.hip dense_hi_select_fuse_kernel): the fp16 image
  // of X, made on first use (not while a stream is capturing) and dropped by add(); the per-query bounds; how many queries
  // re-scored their whole row inside the second pass
  amdr_dense_small_t* small = nullptr;
  bool small_failed = false;
  DevBuf small_eps, small_fb;
  // optional HIP-event ring bracketing the scan kernel alone (bench.py roofline)
  std::vector<hipEvent_t> prof_ev;
  int prof_used = 0;
  bool prof_on = false;
};

namespace {

struct ScanPlan {
  int nq_per_block;  // NQ template
  int ch;
  int cap;
  long rows_per_block;
  int grid_x, grid_y;
  size_t lds;
  size_t part_bytes;
};

int make_plan(int64_t n, int d, int nq, int k, ScanPlan* p) {
  p->ch = ceil_div(d, 256);
  p->cap = topk_cap(k);
  // queries per pass: as many as fit 64 KiB of LDS staging and the register budget
  int nqb = 8;
  while (nqb > 1 && (size_t)kWaves * nqb * p->cap * sizeof(C32) > 60 * 1024) nqb >>= 1;
  if (p->ch >= 4 && nqb > 4) nqb = 4;  // d = 1024: 8 query vectors would spill
  while (nqb > 1 && nqb / 2 >= nq) nqb >>= 1;
  p->nq_per_block = nqb;
  const int U = (nqb <= 2) ? 4 : 2;
  // Row slabs: enough blocks to fill 256 CUs several times over, but never thinner than a
  // few iterations per wave — the per-block top-k finalisation is a fixed cost per slab,
  // so when there are already many query groups (grid_y) the slabs get fatter instead.
  p->grid_y = ceil_div(nq, nqb);
  long min_rows = (long)kWaves * U * 4;
  long want_blocks = 256L * 8;
  long gx = (want_blocks + p->grid_y - 1) / p->grid_y;
  long gx_max = (n + min_rows - 1) / min_rows;
  if (gx > gx_max) gx = gx_max;
  if (gx < 1) gx = 1;
  p->rows_per_block = (n + gx - 1) / gx;
  // round the slab to a multiple of the block's row stride so waves stay aligned
  long stride = (long)kWaves * U;
  p->rows_per_block = ((p->rows_per_block + stride - 1) / stride) * stride;
  if (p->rows_per_block < stride) p->rows_per_block = stride;  // empty index: keep the divisor non-zero
  p->grid_x = (int)((n + p->rows_per_block - 1) / p->rows_per_block);
  if (p->grid_x < 1) p->grid_x = 1;
  p->grid_y = ceil_div(nq, nqb);
  p->lds = (size_t)kWaves * nqb * p->cap * sizeof(C32) + kWaves * sizeof(int);
  p->part_bytes = (size_t)p->grid_x * nq * k * sizeof(C32);
  return AMDR_OK;
}

template <int NQ, int CH>
void launch_scan(const ScanPlan& p, const amdr_dense* h, const float* Q, int nq, int k, C32* part, float* fs,
                 int64_t* fi, hipStream_t st) {
  constexpr int U = (NQ <= 2) ? 4 : 2;
  if (dense_stream_nontemporal((long)h->n, h->d))
    hipLaunchKernelGGL((dense_scan_topk_kernel<NQ, CH, U, true>), dim3(p.grid_x, p.grid_y), dim3(256), p.lds, st, h->X,
                       (long)h->n, h->d, Q, nq, k, p.cap, p.rows_per_block, part, fs, (long long*)fi);
  else
    hipLaunchKernelGGL((dense_scan_topk_kernel<NQ, CH, U, false>), dim3(p.grid_x, p.grid_y), dim3(256), p.lds, st, h->X,
                       (long)h->n, h->d, Q, nq, k, p.cap, p.rows_per_block, part, fs, (long long*)fi);
}

template <int NQ>
int launch_scan_ch(const ScanPlan& p, const amdr_dense* h, const float* Q, int nq, int k, C32* part, float* fs,
                   int64_t* fi, hipStream_t st) {
  switch (p.ch) {
    case 1: launch_scan<NQ, 1>(p, h, Q, nq, k, part, fs, fi, st); break;
    case 2: launch_scan<NQ, 2>(p, h, Q, nq, k, part, fs, fi, st); break;
    case 3: launch_scan<NQ, 3>(p, h, Q, nq, k, part, fs, fi, st); break;
    case 4: launch_scan<NQ, 4>(p, h, Q, nq, k, part, fs, fi, st); break;
    default: return fail(AMDR_EINVAL, "dense: unsupported dim %d", h->d);
  }
  return AMDR_OK;
}

// Batches of >= kBatchedMin queries take the 32-query-tile MFMA path (dense_mfma.hip);
// the score matrix workspace is bounded, so very large batches go in chunks of queries.
constexpr int kBatchedMin = 5;  // measured: from 5 queries up one MFMA tile pass beats the 8-query GEMV pass
constexpr int64_t kRowWavesMax = 16384;  // rows up to which 1-4 queries take one wave per (query, row)
constexpr size_t kScoreBytesMax = (size_t)4 << 30;

int batched_chunk(const amdr_dense* h, int nq) {
  size_t per_q = (size_t)h->n * sizeof(float);
  long c = (long)(kScoreBytesMax / (per_q ? per_q : 1));
  c = (c / 32) * 32;
  if (c < 32) c = 32;
  return nq < c ? nq : (int)c;
}

// Slab-list bytes of one batched search: the full chunk AND the remainder pass (planned for its own size:
// slabs(m) * m is not monotone in m, so a shorter pass can need more).
size_t batched_part_need(const amdr_dense* h, int nq, int k) {
  const int chunk = batched_chunk(h, nq);
  DenseMfmaPlan p;
  dense_mfma_plan((long)h->n, h->d, chunk, k, &p);
  size_t need = p.part_bytes;
  if (nq % chunk) {
    dense_mfma_plan((long)h->n, h->d, nq % chunk, k, &p);
    need = p.part_bytes > need ? p.part_bytes : need;
  }
  return need;
}

// Two-pass form: scores S[q][row] (fp32-MFMA tiles for batches, one wave per (query, row) for the
// 1-4 query call on a short corpus), then slab top-k (+ merge when there are several slabs).
// The two-pass form of a LONG batch on a SHORT corpus: approximate scores on the fp16 matrix instructions (16 x the exact
// form's rate), then per query the rows inside a proven margin of its k-th best re-scored exactly.  From
// 4 096 queries per launch (below, the first pass's fixed cost eats the gain: 1 168 queries 28 us against 26 for the whole
// exact search), one slab of <= 1 024 rows, d a multiple of 128, k (+ the BM25 depth when fused) <= 32.
// AMDR_DENSE_SMALL_HI=0 pins the exact form, AMDR_DENSE_SMALL_HI_MIN the batch size it starts at.
bool small_hi_shape(const amdr_dense* h, int m, int k, int kb) {
  const char* e = getenv("AMDR_DENSE_SMALL_HI");
  if (e && e[0] == '0') return false;
  const char* mn = getenv("AMDR_DENSE_SMALL_HI_MIN");
  const int m_min = mn && atoi(mn) > 0 ? atoi(mn) : 4096;
  // depth <= 12: the second pass finds its candidates with the pair selector's 32 slots per query; from k ~ 14 up those
  // overflow on most queries and the query re-scores its whole row (37 376 queries on 1 024 x 768: k = 12 230 against 517 us
  // for the exact form, k = 14 465 against 536, k = 20 2 099 against 619)
  return m >= m_min && h->n >= 1 && h->n <= 1024 && h->d >= 128 && h->d <= 1024 && h->d % 128 == 0 && k >= 1 && k <= 12 &&
         k + kb <= 32;
}
// the image and the workspaces of that form; false: not available now (creation failed before, non-finite matrix, or a
// stream is capturing and nothing was reserved) — the caller takes the exact form
bool small_hi_ready(amdr_dense* h, int m, hipStream_t st) {
  if (h->small_failed) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (!h->small) {
    if (capturing) return false;
    if (dense_small_create_from(h->device, h->X, h->n, h->d, &h->small) != AMDR_OK || !dense_small_usable(h->small)) {
      h->small_failed = true;  // (a matrix the fp16 scale range cannot hold stays on the exact form)
      return false;
    }
  }
  const size_t need = (size_t)m * sizeof(float);
  if (capturing && (h->small_eps.cap < need || !h->small_fb.p)) return false;
  if (h->small_eps.ensure(need) != AMDR_OK) return false;
  if (!h->small_fb.p) {
    if (h->small_fb.ensure(sizeof(unsigned int)) != AMDR_OK) return false;
    (void)hipMemset(h->small_fb.p, 0, sizeof(unsigned int));
  }
  return capturing ? true : dense_small_reserve(h->small, m) == AMDR_OK;
}

int run_search_batched(amdr_dense* h, int ws, const float* Q_dev, int nq, int k, float* scores_dev, int64_t* ids_dev,
                       hipStream_t st, bool row_waves = false, const FuseTail* tail = nullptr) {
  DevBuf& smat = h->smat[ws];
  DevBuf& partb = h->part[ws];
  const int chunk = batched_chunk(h, nq);
  DenseMfmaPlan p;
  dense_mfma_plan((long)h->n, h->d, chunk, k, &p);
  int rc = smat.ensure(p.s_bytes);
  if (!rc) rc = partb.ensure(batched_part_need(h, nq, k));
  if (rc) return rc;
  for (int q0 = 0; q0 < nq; q0 += chunk) {
    const int m = nq - q0 < chunk ? nq - q0 : chunk;
    if (m != chunk) dense_mfma_plan((long)h->n, h->d, m, k, &p);
    const bool prof = h->prof_on && (size_t)(h->prof_used + 2) <= h->prof_ev.size();
    if (prof) AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
    const int kb_fused = tail ? tail->kb : 0;
    const bool fuse_here = tail && dense_select_fuse_applies((long)h->n, p.slabs, m, k, tail->kb);
    if (!row_waves && p.slabs == 1 && (fuse_here || !tail) && small_hi_shape(h, m, k, kb_fused) && small_hi_ready(h, m, st)) {
      rc = amdr_dense_small_approx_device(h->small, Q_dev + (size_t)q0 * h->d, m, smat.as<float>(), p.ld,
                                          h->small_eps.as<float>(), st);
      if (rc) return rc;
      if (prof) {
        AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
        h->prof_used += 2;
      }
      if ((rc = dense_hi_select_launch(tail, q0, smat.as<float>(), p.ld, (long)h->n, m, k, h->X, Q_dev + (size_t)q0 * h->d, h->d,
                                       h->small_eps.as<float>(), scores_dev + (size_t)q0 * k, ids_dev + (size_t)q0 * k,
                                       h->small_fb.as<unsigned int>(), st)))
        return rc;
      continue;
    }
    if (row_waves) {
      hipLaunchKernelGGL(dense_all_scores_kernel, dim3(ceil_div((long)m * h->n, kWaves)), dim3(256), 0, st, h->X,
                         (long)h->n, h->d, Q_dev + (size_t)q0 * h->d, m, p.ld, smat.as<float>());
      AMDR_HIP(hipGetLastError());
    } else if (dense_panel_supported((long)h->n, h->d, m)) {
      DensePanelPlan pp;
      dense_panel_plan((long)h->n, h->d, m, &pp);
      rc = dense_panel_launch_scores(pp, h->X, (long)h->n, h->d, Q_dev + (size_t)q0 * h->d, m, p.ld,
                                     smat.as<float>(), st);
      if (rc) return rc;
    } else {
      rc = dense_mfma_launch_scores(p, h->X, (long)h->n, h->d, Q_dev + (size_t)q0 * h->d, m, smat.as<float>(), st);
      if (rc) return rc;
    }
    if (prof) {
      AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
      h->prof_used += 2;
    }
    const bool direct = p.slabs == 1;  // one slab: its list is the answer, no merge launch
    if (tail && dense_select_fuse_applies((long)h->n, p.slabs, m, k, tail->kb)) {
      // ranking of the rows and the fusion with the BM25 lists in one kernel (fuse.hip dense_select_fuse_kernel)
      if ((rc = dense_select_fuse_launch(*tail, q0, smat.as<float>(), p.ld, (long)h->n, m, k, p.cap,
                                         scores_dev + (size_t)q0 * k, ids_dev + (size_t)q0 * k, st)))
        return rc;
      continue;
    }
    rc = dense_mfma_launch_topk(p, smat.as<float>(), (long)h->n, m, k, partb.p,
                                direct ? scores_dev + (size_t)q0 * k : nullptr,
                                direct ? ids_dev + (size_t)q0 * k : nullptr, st);
    if (rc) return rc;
    if (!direct) {
      size_t lds = (size_t)kWaves * p.cap * sizeof(C32) + kWaves * sizeof(int);
      hipLaunchKernelGGL(dense_merge_kernel, dim3(m), dim3(256), lds, st, partb.as<C32>(), p.slabs, m, k, p.cap,
                         scores_dev + (size_t)q0 * k, (long long*)ids_dev + (size_t)q0 * k, (const int*)nullptr);
      AMDR_HIP(hipGetLastError());
    }
    if (tail && (rc = dense_fuse_plain_launch(*tail, q0, m, k, scores_dev + (size_t)q0 * k, ids_dev + (size_t)q0 * k, st)))
      return rc;
  }
  return AMDR_OK;
}

// Two-level top-k for batches on a matrix far larger than the caches (the 10 M-row scans): the score matrix of the
// plain two-pass form is 4 B x queries per row — 1.28 GB per 32 queries on 10 M rows, written and read back once, and
// its stores interleave with the read stream at the HBM (timing-only build without them: 5.80 -> 5.10 ms).  Here:
//   1. the tile kernel keeps, per 32-row tile and query, only the MAXIMUM  (n/32 x queries floats: 40 MB);
//   2. top-k of each query's tile maxima -> k candidate tiles; their union, sorted, without duplicates;
//   3. the tile kernel re-scores the candidate tiles (same loads, same MFMA k order: the same bits as a full pass);
//   4. top-k of the re-scored columns, columns -> row ids.
// Exact: at most k - 1 tiles hold a score above a query's k-th best s_k, so the k-th largest tile maximum T <= s_k
// and every tile that holds one of the top k has a maximum >= T; among tiles AT T the lower tile ids are kept, which
// is where the lower row ids of equal scores live.  queries x k <= 8 192 candidate tiles per pass.
bool two_level_applies(const amdr_dense* h, int nq, int k) {
  const char* e = getenv("AMDR_DENSE_TWO_LEVEL");
  if (e && e[0] == '0') return false;
  if (!(nq >= kBatchedMin && h->n > 0 && dense_mfma_supported(h->d))) return false;
  const long tiles = ((long)h->n + 31) / 32;
  if (e && e[0] == '1') return tiles >= 2L * k;  // pinned on (tests): any matrix with enough tiles
  if (dense_panel_supported((long)h->n, h->d, batched_chunk(h, nq))) return false;  // >= 96 queries: the panel kernel
  return dense_stream_nontemporal((long)h->n, h->d) && tiles >= 64L * k;
}
constexpr int kTwoLevelTilesMax = 8192;  // candidate tiles per pass (queries x k): one wave sorts them in 64 KiB of LDS
int two_level_chunk(int nq, int k) {
  int c = (kTwoLevelTilesMax / k) / 32 * 32;  // >= 32 for every k <= AMDR_MAX_K = 256
  if (c < 32) c = 32;
  if (c > 96) c = 96;
  return nq < c ? nq : c;
}

// The fp16 first pass (dense_hi.hip): the tile maxima of step 1 come from v_mfma_f32_32x32x16_f16 on fp16 roundings of
// both operands — 64 queries per scan instead of 32, the scan bound by HBM alone.  Approximate maxima a(t) lie within
// eps_q of the exact ones (dense_hi_check_kernel states the bound), so the candidate set is widened: the kc = k +
// max(k, 22) + 1 tiles with the largest a(t) are re-scored, and the answer is exact if the kc-th largest a(t) lies below
// T_k - 2 eps_q (T_k = the k-th largest): the k tiles on top have exact maxima >= T_k - eps, hence s_k >= T_k - eps,
// and a tile holding a row >= s_k has a(t) >= s_k - eps >= T_k - 2 eps — it is among the first kc - 1.  Steps 3-4 are
// the unchanged exact kernels: same ids, same score bits.  A query the bound does not separate raises a device flag;
// the exact chain is enqueued behind, every launch gated on that flag (no host round trip), and rewrites the batch.
// Extra candidates: the tiles expected within 2 eps below the cut grow with k (about 0.4 k on unit-norm Gaussian rows)
// and with how tightly the matrix clusters around a query's best rows, which only the data knows: three widths.
constexpr int kHiLevels = 3;
constexpr int kHiExtra[kHiLevels] = {22, 54, 96};
int hi_kc(int k, int level) { return k + (k > kHiExtra[level] ? k : kHiExtra[level]) + 1; }
int hi_kc_max(int k) { return hi_kc(k, kHiLevels - 1); }
int hi_level_of(const amdr_dense* h) {
  const char* e = getenv("AMDR_DENSE_HI_LEVEL");  // pins the width (tests, A/B)
  if (e && e[0] >= '0' && e[0] < '0' + kHiLevels) return e[0] - '0';
  return h->hi_level;
}
bool hi_applies(const amdr_dense* h, int nq, int k) {
  const char* e = getenv("AMDR_DENSE_HI");
  if (e && e[0] == '0') return false;
  const char* e2 = getenv("AMDR_DENSE_TWO_LEVEL");
  if (e2 && e2[0] == '0') return false;
  if (!(h->hi_ok && nq >= kBatchedMin && h->n > 0 && dense_mfma_supported(h->d))) return false;
  if (hi_kc_max(k) > AMDR_MAX_K) return false;  // k <= 127
  const long tiles = ((long)h->n + 31) / 32;
  if (tiles >= (1l << 26)) return false;  // (query, tile) packed in 32 bits of a candidate entry
  if (e && e[0] == '1') return tiles >= 2L * hi_kc_max(k);  // pinned on (tests)
  return dense_stream_nontemporal((long)h->n, h->d) && tiles >= 64L * hi_kc_max(k);
}
int hi_chunk(const amdr_dense* h, int nq, int k) {  // the same at every level: a handle's passes keep their shape when its level moves
  int c = dense_hi_max_queries(h->d);
  // the candidate union is a bitmap over the tiles (any number of candidates) up to 2^20 tiles; beyond, the one-wave sort
  // and its 8 192-candidate limit (>= 32 queries for every admitted k)
  if (((long)h->n + 31) / 32 > kUniqueBitmapTilesMax && kTwoLevelTilesMax / hi_kc_max(k) < c) c = kTwoLevelTilesMax / hi_kc_max(k);
  return nq < c ? nq : c;
}
// Between searches (host side, no synchronisation: the counters are whatever the last completed copy-back left).  One
// unresolved query sends its whole pass through the exact chain as well (+2 scans for a 1-scan pass), so what is
// counted is PASSES whose flag went up: more than 10 % of >= 4 passes at this width -> the next width (+3 % per pass);
// at the widest, more than half -> the exact passes alone are cheaper (1 + 2 f > 2).
bool hi_tail2();
void hi_adapt(amdr_dense* h) {
  if (!h->hi_host || h->hi_off || getenv("AMDR_DENSE_HI_LEVEL")) return;
  // the counters are whatever the last COMPLETED copy-back left: the pinned words are read only after the event behind
  // their copy has been reached (round 3 read them while a copy could still be in flight)
  if (h->hi_copy_pending) {
    if (hipEventQuery(h->hi_ev) != hipSuccess) return;  // still on its way: adapt at the next search
    h->hi_copy_pending = false;
    for (int i = 0; i < 3; ++i) h->hi_seen[i] = h->hi_host[i];
  }
  const unsigned int f = h->hi_seen[1];
  // passes: counted on the device next to the flags under the round-4 tail (a hipGraph replay bumps both; the host's
  // own count would not see replays), on the host under the round-3 tail
  const int64_t passes = hi_tail2() ? (int64_t)h->hi_seen[2] : h->hi_passes;
  const int64_t p = passes - h->lvl_p0;
  const int64_t bad = (int64_t)(f - h->lvl_f0);
  if (p < 4) return;
  bool move = false;
  if (h->hi_level + 1 < kHiLevels) {
    move = bad * 10 > p;
    if (move) ++h->hi_level;
  } else {
    const char* e = getenv("AMDR_DENSE_HI");
    move = bad * 2 > p && !(e && e[0] == '1');  // this matrix is not for the fp16 pass
    if (move) h->hi_off = true;
  }
  if (move || p >= (1 << 16)) {  // a new window
    h->lvl_p0 = passes;
    h->lvl_f0 = f;
  }
}
// AMDR_DENSE_HI_TAIL=0 pins the round-3 tail (flat candidate list, ~18 launches per 64-query pass): A/B, tests
bool hi_tail2() {
  const char* e = getenv("AMDR_DENSE_HI_TAIL");
  return !(e && e[0] == '0');
}
constexpr int kHi2Tiles = 4;  // query tiles per pass of the round-4 tail: 256 queries (192 at d = 1 024) share one tail
int hi2_chunk(const amdr_dense* h, int nq) {
  const int c = kHi2Tiles * dense_hi_max_queries(h->d);
  return nq < c ? nq : c;
}

// Workspace of one pass of the round-4 tail.  smat: exact tile maxima M [m][ldM] (written only when the flag goes up) |
// re-scored columns S2 [m][32 kc] | sample maxima MT [qtiles][items][64] | per-query candidate lists [m][qcap];
// aux: list [m][kc] | count, unres [m] | tau [m] | qcount [m] (the gate flag sits at the end of aux, as before).
struct Hi2Plan {
  int qtiles, kc;
  long tiles, ldM, ldS2;
  size_t qcap;
  DenseMfmaPlan scan;
  size_t off_S2, off_MT, off_qlist, smat_bytes;
  size_t off_count, off_unres, off_tau, off_qcount, aux_bytes;
};
void hi2_plan(const amdr_dense* h, int m, int k, int kc, Hi2Plan* p) {
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const int qt = dense_hi_max_queries(h->d);
  p->qtiles = (m + qt - 1) / qt;
  p->kc = kc;
  p->tiles = ((long)h->n + 31) / 32;
  dense_mfma_plan((long)h->n, h->d, m, k, &p->scan);
  p->ldM = (p->tiles + 31) / 32 * 32;
  p->scan.ld = p->ldM;
  p->ldS2 = (long)kc * 32;
  p->qcap = dense_hi2_qcap((long)h->n, p->qtiles, kc);
  p->off_S2 = up((size_t)m * p->ldM * sizeof(float));
  p->off_MT = p->off_S2 + up((size_t)m * p->ldS2 * sizeof(float));
  p->off_qlist = p->off_MT + up((size_t)(2048 + 64 * kHi2Tiles) * 64 * sizeof(float));  // [queries][items rounded up to 64]
  p->smat_bytes = p->off_qlist + up((size_t)m * p->qcap * sizeof(C32));
  p->off_count = up((size_t)m * kc * sizeof(int));
  p->off_unres = p->off_count + up((size_t)m * sizeof(int));
  p->off_tau = p->off_unres + up((size_t)m * sizeof(int));
  p->off_qcount = p->off_tau + up((size_t)m * sizeof(float));
  p->aux_bytes = p->off_qcount + up((size_t)m * sizeof(unsigned int));
}

struct TwoLevelPlan {
  DenseMfmaPlan scan, tk1, pass2;  // full scan (mode 1), top-kc over the tile maxima, candidate re-scoring + its top-k
  long tiles, cand_rows;
  size_t m_bytes, s2_bytes, aux_bytes, part_bytes;
  // fp16 first pass only: the sample (tk1 then ranks ITS maxima), the flat candidate list
  long sample_items;
  size_t s2_own, mt_bytes, cand_entries;
};
// kc = candidate tiles per query: k in the exact form, hi_kc(k, level) behind the fp16 first pass
void two_level_plan(const amdr_dense* h, int m, int k, int kc, TwoLevelPlan* t) {
  const bool hi = kc != k;
  t->tiles = ((long)h->n + 31) / 32;
  t->sample_items = hi ? dense_hi_sample_items((long)h->n) : 0;
  if (!hi) dense_mfma_plan((long)h->n, h->d, m, k, &t->scan);  // the exact first pass only
  dense_mfma_plan(hi ? t->sample_items : t->tiles, h->d, m, kc, &t->tk1);  // only its top-k half is used: columns = tiles
  // exact form: every query of the pass against the UNION of their candidate tiles (one list); behind the fp16 pass a
  // query against its own kc tiles (mode 3 of the scores kernel: block row = query, <= 16 tiles per block)
  t->cand_rows = hi ? (long)kc * 32 : (long)m * kc * 32;
  dense_mfma_plan(t->cand_rows, h->d, m, k, &t->pass2);
  if (hi) {
    // a (query, tile) pass is 11.7 us of fp32 matrix time on one SIMD and the check drops the tiles below a query's cut
    // (~14 of 33 stay): four tiles per block = one per SIMD, blocks beyond a query's list return before staging anything
    t->pass2.rows_per_block = 4 * 32;
    t->pass2.grid_x = (int)((t->cand_rows + t->pass2.rows_per_block - 1) / t->pass2.rows_per_block);
    t->pass2.grid_y = m;
  }
  t->m_bytes = ((size_t)m * t->tk1.ld * sizeof(float) + 255) / 256 * 256;
  t->s2_own = ((size_t)m * t->pass2.ld * sizeof(float) + 255) / 256 * 256;
  t->mt_bytes = hi ? (dense_hi_mt_bytes((long)h->n) + 255) / 256 * 256 : 0;
  t->cand_entries = hi ? dense_hi_cand_entries((long)h->n, m, kc) : 0;
  t->s2_bytes = t->s2_own + t->mt_bytes + t->cand_entries * sizeof(C32);  // S2 | sample maxima | candidate list
  // tile ids + maxima, union list + count; behind the fp16 pass also the sample's top-kc (ids + maxima)
  t->aux_bytes = (size_t)m * kc * (sizeof(int64_t) + sizeof(float)) * (hi ? 2 : 1) + (size_t)(m * kc + 64) * sizeof(int) + 256;
  t->part_bytes = t->tk1.part_bytes > t->pass2.part_bytes ? t->tk1.part_bytes : t->pass2.part_bytes;
  if (hi && dense_hi_cand_part_bytes(m, kc) > t->part_bytes) t->part_bytes = dense_hi_cand_part_bytes(m, kc);
}
// Workspace for one search of nq queries at depth k: the maximum over every chunk size the pass loop will use — the
// full chunk AND the remainder (a smaller chunk can need MORE slab-list space: slabs(m) * m is not monotone in m).
// With `all` (amdr_dense_reserve: calls within (nq_max, k_max) must allocate nothing) the maximum over every batch
// size <= nq and every depth <= k that takes this path — a smaller k takes more queries per pass (a larger matrix of
// tile maxima) and the path's own applicability test depends on k.
struct TwoLevelNeed {
  size_t smat = 0, part = 0, aux = 0;
  void add(const TwoLevelPlan& t) {
    smat = t.m_bytes + t.s2_bytes > smat ? t.m_bytes + t.s2_bytes : smat;
    part = t.part_bytes > part ? t.part_bytes : part;
    aux = t.aux_bytes + 256 > aux ? t.aux_bytes + 256 : aux;  // + the gate flag behind the lists
  }
};
void two_level_need_exact(const amdr_dense* h, int nq, int k, TwoLevelNeed* need) {
  const int chunk = two_level_chunk(nq, k);
  TwoLevelPlan t;
  two_level_plan(h, chunk, k, k, &t);
  need->add(t);
  if (nq % chunk) {
    two_level_plan(h, nq % chunk, k, k, &t);
    need->add(t);
  }
}
void hi2_need(const amdr_dense* h, int nq, int k, TwoLevelNeed* need) {  // monotone in the pass size and in kc
  Hi2Plan p;
  hi2_plan(h, hi2_chunk(h, nq), k, hi_kc_max(k), &p);
  need->smat = p.smat_bytes > need->smat ? p.smat_bytes : need->smat;
  need->aux = p.aux_bytes + 256 > need->aux ? p.aux_bytes + 256 : need->aux;
}
void two_level_need(const amdr_dense* h, int nq, int k, TwoLevelNeed* need) {
  if (!hi_applies(h, nq, k)) return two_level_need_exact(h, nq, k, need);
  if (hi_tail2()) hi2_need(h, nq, k, need);
  const int chunk = hi_chunk(h, nq, k);
  TwoLevelPlan t;
  for (int m : {chunk, nq % chunk}) {
    if (m == 0) continue;
    for (int l = 0; l < kHiLevels; ++l) {  // the level can move between searches: reserve for all three
      two_level_plan(h, m, k, hi_kc(k, l), &t);
      need->add(t);
    }
    two_level_need_exact(h, m, k, need);  // the gated exact chain of the same pass
  }
}
int two_level_ensure(amdr_dense* h, int ws, int nq, int k, bool all = false) {
  TwoLevelNeed need;
  if (!all) {
    two_level_need(h, nq, k, &need);
  } else {
    for (int kk = 1; kk <= k; ++kk) {
      // a call's path is chosen on its whole batch; its passes (full chunks and a remainder of ANY size) then
      // all run the two-level form: cover every m a pass can have
      const bool hi = hi_applies(h, nq, kk);
      if (!hi && !two_level_applies(h, nq < 95 ? nq : 95, kk) && !two_level_applies(h, nq, kk)) continue;
      if (hi && hi_tail2()) hi2_need(h, nq, kk, &need);
      TwoLevelPlan t;
      if (hi)
        for (int m = 1; m <= hi_chunk(h, nq, kk); ++m)
          for (int l = 0; l < kHiLevels; ++l) {
            two_level_plan(h, m, kk, hi_kc(kk, l), &t);
            need.add(t);
          }
      const int cmax = two_level_chunk(nq, kk);
      for (int m = 1; m <= cmax; ++m) {
        two_level_plan(h, m, kk, kk, &t);
        need.add(t);
      }
    }
  }
  int rc = need.smat ? h->smat[ws].ensure(need.smat) : AMDR_OK;
  if (!rc && need.part) rc = h->part[ws].ensure(need.part);
  if (!rc && need.aux) rc = h->aux[ws].ensure(need.aux);
  return rc;
}

// one top-k pass over a [m][ld] score matrix with `cols` valid columns (slab lists + merge, or direct)
int topk_pass(const DenseMfmaPlan& p, const float* S, long cols, int m, int k, DevBuf& partb, float* out_scores,
              int64_t* out_ids, hipStream_t st, const int* gate = nullptr) {
  const bool direct = p.slabs == 1;
  int rc = dense_mfma_launch_topk(p, S, cols, m, k, partb.p, direct ? out_scores : nullptr, direct ? out_ids : nullptr, st,
                                  gate);
  if (rc) return rc;
  if (!direct) {
    size_t lds = (size_t)kWaves * p.cap * sizeof(C32) + kWaves * sizeof(int);
    hipLaunchKernelGGL(dense_merge_kernel, dim3(m), dim3(256), lds, st, partb.as<C32>(), p.slabs, m, k, p.cap, out_scores,
                       (long long*)out_ids, gate);
    AMDR_HIP(hipGetLastError());
  }
  return AMDR_OK;
}

// One pass of <= chunk queries.  kc_hi > 0: the fp16 first pass with kc_hi candidate tiles per query, its check raises
// *flag; otherwise the exact first pass, every launch gated on *gate when given.
int two_level_pass(amdr_dense* h, int ws, const float* Qc, int m, int k, int kc_hi, float* out_scores, int64_t* out_ids,
                   hipStream_t st, int* flag, const int* gate) {
  const bool hi = kc_hi > 0;
  const int kc = hi ? kc_hi : k;
  TwoLevelPlan t;
  two_level_plan(h, m, k, kc, &t);
  float* M = h->smat[ws].as<float>();
  unsigned char* s2p = reinterpret_cast<unsigned char*>(h->smat[ws].p) + t.m_bytes;
  float* S2 = reinterpret_cast<float*>(s2p);
  unsigned char* aux = reinterpret_cast<unsigned char*>(h->aux[ws].p);
  int64_t* tile_ids = reinterpret_cast<int64_t*>(aux);
  float* tile_max = reinterpret_cast<float*>(aux + (size_t)m * kc * sizeof(int64_t));
  int* list = reinterpret_cast<int*>(aux + (size_t)m * kc * (sizeof(int64_t) + sizeof(float)));
  int* count = list + (size_t)m * kc;
  int rc;
  const bool prof = !gate && h->prof_on && (size_t)(h->prof_used + 2) <= h->prof_ev.size();
  if (hi) {
    // 1a. maxima of a strided sample of the tiles -> per query the kc-th best = the threshold of the full scan
    float* MT = reinterpret_cast<float*>(s2p + t.s2_own);
    void* cand = s2p + t.s2_own + t.mt_bytes;
    int64_t* ts_ids = reinterpret_cast<int64_t*>(count + 64);
    float* ts_max = reinterpret_cast<float*>(ts_ids + (size_t)m * kc);
    unsigned int* total = reinterpret_cast<unsigned int*>(flag) + 16;
    if ((rc = dense_hi_launch_sample(h->X, (long)h->n, h->d, Qc, m, MT, st, h->x_scale))) return rc;
    if ((rc = dense_hi_launch_transpose(MT, t.sample_items, m, t.tk1.ld, M, st))) return rc;
    if ((rc = topk_pass(t.tk1, M, t.sample_items, m, kc, h->part[ws], ts_max, ts_ids, st))) return rc;
    AMDR_HIP(hipMemsetAsync(total, 0, sizeof(unsigned int), st));
    // 1b. the scan (the launch the profiling events bracket): maxima that reach the threshold -> flat candidate list
    if (prof) AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
    if ((rc = dense_hi_launch_emit(h->X, (long)h->n, h->d, Qc, m, ts_max + (kc - 1), kc, cand, total, t.cand_entries, st,
                                   h->x_scale)))
      return rc;
    if (prof) {
      AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
      h->prof_used += 2;
    }
    // 2. kc candidate tiles per query; is the cut wide enough?
    int nparts = 0;
    if ((rc = dense_hi_launch_cand_topk(cand, total, t.cand_entries, m, kc, h->part[ws].p, &nparts, st))) return rc;
    {
      const int mcap = topk_cap(kc);
      const size_t lds = (size_t)kWaves * mcap * sizeof(C32) + kWaves * sizeof(int);
      hipLaunchKernelGGL(dense_merge_kernel, dim3(m), dim3(256), lds, st, h->part[ws].as<C32>(), nparts, m, kc, mcap, tile_max,
                         (long long*)tile_ids, (const int*)nullptr);
      AMDR_HIP(hipGetLastError());
    }
    h->hi_queries += m;
    h->hi_passes += 1;
    if ((rc = dense_hi_launch_check(tile_max, tile_ids, m, kc, k, Qc, h->d, h->row_norm_max, h->x_scale, t.tiles, total,
                                    t.cand_entries, flag, h->stats.as<unsigned int>() + 2, st)))
      return rc;
  } else {
    // 1. tile maxima (the scan: this is the launch the profiling events bracket)
    if (prof) AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
    DenseMfmaPlan scan = t.scan;
    scan.ld = t.tk1.ld;
    if ((rc = dense_mfma_launch_scores(scan, h->X, (long)h->n, h->d, Qc, m, M, st, 1, nullptr, nullptr, 0, gate))) return rc;
    if (prof) {
      AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
      h->prof_used += 2;
    }
    // 2. k candidate tiles per query
    if ((rc = topk_pass(t.tk1, M, t.tiles, m, kc, h->part[ws], tile_max, tile_ids, st, gate))) return rc;
  }
  if (hi) {
    // 3. every query's own candidate tiles, ascending; their exact scores; 4. top-k, columns -> row ids
    if ((rc = dense_tiles_sort_per_query_launch(tile_ids, m, kc, list, count, st))) return rc;
    if ((rc = dense_mfma_launch_scores(t.pass2, h->X, t.cand_rows, h->d, Qc, m, S2, st, 3, list, count, (long)h->n, nullptr, kc)))
      return rc;
    if ((rc = topk_pass(t.pass2, S2, t.cand_rows, m, k, h->part[ws], out_scores, out_ids, st))) return rc;
    return dense_tiles_remap_launch(out_ids, m * k, list, count, (long)h->n, st, nullptr, k, kc);
  }
  // ... their sorted union
  if ((rc = dense_tiles_unique_launch(tile_ids, m * kc, t.tiles, list, count, st, gate))) return rc;
  // 3. exact scores of the candidate tiles' rows
  if ((rc = dense_mfma_launch_scores(t.pass2, h->X, t.cand_rows, h->d, Qc, m, S2, st, 2, list, count, (long)h->n, gate)))
    return rc;
  // 4. top-k of the candidates, columns -> row ids
  if ((rc = topk_pass(t.pass2, S2, t.cand_rows, m, k, h->part[ws], out_scores, out_ids, st, gate))) return rc;
  return dense_tiles_remap_launch(out_ids, m * k, list, count, (long)h->n, st, gate);
}

// One pass (<= 4 query tiles) of the round-4 tail: see dense_hi.hip.  Launches: sample, tau, one scan per query tile,
// select, [gated: exact tile maxima, exact select], re-scoring, final top-k.
int hi2_pass(amdr_dense* h, int ws, const float* Qc, int m, int k, int kc, float* out_scores, int64_t* out_ids, hipStream_t st,
             int* flag) {
  Hi2Plan p;
  hi2_plan(h, m, k, kc, &p);
  unsigned char* sm = reinterpret_cast<unsigned char*>(h->smat[ws].p);
  unsigned char* ax = reinterpret_cast<unsigned char*>(h->aux[ws].p);
  float* M = reinterpret_cast<float*>(sm);
  float* S2 = reinterpret_cast<float*>(sm + p.off_S2);
  float* MT = reinterpret_cast<float*>(sm + p.off_MT);
  C32* qlist = reinterpret_cast<C32*>(sm + p.off_qlist);
  int* list = reinterpret_cast<int*>(ax);
  int* count = reinterpret_cast<int*>(ax + p.off_count);
  int* unres = reinterpret_cast<int*>(ax + p.off_unres);
  float* tau = reinterpret_cast<float*>(ax + p.off_tau);
  unsigned int* qcount = reinterpret_cast<unsigned int*>(ax + p.off_qcount);
  unsigned int* stats = h->stats.as<unsigned int>();
  int rc;
  if ((rc = dense_hi2_launch_sample(h->X, (long)h->n, h->d, Qc, m, p.qtiles, MT, st, h->x_scale))) return rc;
  if ((rc = dense_hi2_launch_tau(MT, (long)h->n, h->d, m, p.qtiles, kc, tau, qcount, flag, stats, st))) return rc;
  {  // the scan: ONE launch over all query tiles of the pass (the launch the profiling events bracket);
     // AMDR_DENSE_HI_SCANS=split: one launch per query tile (A/B, tests)
    const char* sp = getenv("AMDR_DENSE_HI_SCANS");
    const bool split = sp && sp[0] == 's';
    const int qt = dense_hi_max_queries(h->d);
    for (int y = 0; y < (split ? p.qtiles : 1); ++y) {
      const int q0 = y * qt, mq = split ? (m - q0 < qt ? m - q0 : qt) : m;
      const bool prof = h->prof_on && (size_t)(h->prof_used + 2) <= h->prof_ev.size();
      if (prof) AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
      if ((rc = dense_hi2_launch_emit(h->X, (long)h->n, h->d, Qc + (size_t)q0 * h->d, mq, tau + q0, qlist + (size_t)q0 * p.qcap,
                                      qcount + q0, p.qcap, st, h->x_scale, split ? 1 : p.qtiles)))
        return rc;
      if (prof) {
        AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
        h->prof_used += 2;
      }
    }
  }
  h->hi_queries += m;
  h->hi_passes += p.qtiles;
  if ((rc = dense_hi2_launch_select(qlist, qcount, p.qcap, m, kc, k, Qc, h->d, h->row_norm_max, h->x_scale, p.tiles, list,
                                    count, unres, flag, stats + 2, st)))
    return rc;
  // behind the flag: the exact first pass of the batch and, for the queries the bound did not resolve, their k tiles
  if ((rc = dense_mfma_launch_scores(p.scan, h->X, (long)h->n, h->d, Qc, m, M, st, 1, nullptr, nullptr, 0, flag))) return rc;
  if ((rc = dense_hi2_launch_exact_select(M, p.ldM, p.tiles, m, k, kc, list, count, unres, flag, st))) return rc;
  if ((rc = dense_rescore_tiles_launch(h->X, (long)h->n, h->d, Qc, m, list, count, kc, kc, p.ldS2, S2, st))) return rc;
  return dense_final_topk_launch(S2, p.ldS2, list, count, kc, kc, (long)h->n, m, k, out_scores, out_ids, st);
}

int run_search_two_level(amdr_dense* h, int ws, const float* Q_dev, int nq, int k, float* scores_dev, int64_t* ids_dev,
                         hipStream_t st) {
  int rc = two_level_ensure(h, ws, nq, k);
  if (rc) return rc;
  const bool hi = hi_applies(h, nq, k);
  hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap_st);
  const bool capturing = cap_st != hipStreamCaptureStatusNone;
  if (hi && !capturing) hi_adapt(h);
  const bool tail2 = hi && hi_tail2() && !h->hi_off;
  const int chunk = tail2 ? hi2_chunk(h, nq) : hi ? hi_chunk(h, nq, k) : two_level_chunk(nq, k);
  const int kc_hi = hi && !h->hi_off ? hi_kc(k, hi_level_of(h)) : 0;
  // the gate flag: behind the largest list layout of this call (two_level_ensure sized aux for it)
  int* flag = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(h->aux[ws].p) + h->aux[ws].cap - 256);
  for (int q0 = 0; q0 < nq; q0 += chunk) {
    const int m = nq - q0 < chunk ? nq - q0 : chunk;
    const float* Qc = Q_dev + (size_t)q0 * h->d;
    float* os = scores_dev + (size_t)q0 * k;
    int64_t* oi = ids_dev + (size_t)q0 * k;
    if (!hi) {
      if ((rc = two_level_pass(h, ws, Qc, m, k, 0, os, oi, st, nullptr, nullptr))) return rc;
      continue;
    }
    if (tail2) {  // (the flag is reset by the pass's own tau kernel)
      if ((rc = hi2_pass(h, ws, Qc, m, k, kc_hi, os, oi, st, flag))) return rc;
      continue;
    }
    if (kc_hi) {
      AMDR_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
      if ((rc = two_level_pass(h, ws, Qc, m, k, kc_hi, os, oi, st, flag, nullptr))) return rc;
    }
    // the exact chain holds 32-query tiles; behind the fp16 pass it runs only if the flag was raised — and
    // unconditionally, in the same pass shapes (same workspace), on a handle that gave the fp16 pass up
    const int ec = two_level_chunk(m, k);
    for (int e0 = 0; e0 < m; e0 += ec) {
      const int em = m - e0 < ec ? m - e0 : ec;
      if ((rc = two_level_pass(h, ws, Qc + (size_t)e0 * h->d, em, k, 0, os + (size_t)e0 * k, oi + (size_t)e0 * k, st,
                               nullptr, kc_hi ? flag : nullptr)))
        return rc;
    }
  }
  if (kc_hi && h->hi_host && !capturing && !h->hi_copy_pending) {  // what hi_adapt reads before a later search
    AMDR_HIP(hipMemcpyAsync(h->hi_host, h->stats.as<unsigned int>() + 2, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    AMDR_HIP(hipEventRecord(h->hi_ev, st));
    h->hi_copy_pending = true;
  }
  return AMDR_OK;
}

int run_search(amdr_dense* h, int ws, const float* Q_dev, int nq, int k, float* scores_dev, int64_t* ids_dev,
               hipStream_t st) {
  if (hi_applies(h, nq, k) || two_level_applies(h, nq, k))
    return run_search_two_level(h, ws, Q_dev, nq, k, scores_dev, ids_dev, st);
  if (nq >= kBatchedMin && h->n > 0 && dense_mfma_supported(h->d))
    return run_search_batched(h, ws, Q_dev, nq, k, scores_dev, ids_dev, st);
  if (h->n > 0 && h->n <= kRowWavesMax) return run_search_batched(h, ws, Q_dev, nq, k, scores_dev, ids_dev, st, true);
  ScanPlan p;
  make_plan(h->n, h->d, nq, k, &p);
  int rc = h->part[ws].ensure(p.part_bytes);
  if (rc) return rc;
  C32* part = h->part[ws].as<C32>();
  const bool prof = h->prof_on && (size_t)(h->prof_used + 2) <= h->prof_ev.size() && h->n > 0;
  if (prof) AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
  const bool direct = h->n > 0 && p.grid_x == 1;
  float* fs = direct ? scores_dev : nullptr;
  int64_t* fi = direct ? ids_dev : nullptr;
  if (h->n > 0) {
    switch (p.nq_per_block) {
      case 1: rc = launch_scan_ch<1>(p, h, Q_dev, nq, k, part, fs, fi, st); break;
      case 2: rc = launch_scan_ch<2>(p, h, Q_dev, nq, k, part, fs, fi, st); break;
      case 4: rc = launch_scan_ch<4>(p, h, Q_dev, nq, k, part, fs, fi, st); break;
      default: rc = launch_scan_ch<8>(p, h, Q_dev, nq, k, part, fs, fi, st); break;
    }
    if (rc) return rc;
    AMDR_HIP(hipGetLastError());
  }
  if (prof) {
    AMDR_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
    h->prof_used += 2;
  }
  if (direct) return AMDR_OK;
  int nparts = h->n > 0 ? p.grid_x : 0;
  size_t lds = (size_t)kWaves * p.cap * sizeof(C32) + kWaves * sizeof(int);
  hipLaunchKernelGGL(dense_merge_kernel, dim3(nq), dim3(256), lds, st, part, nparts, nq, k, p.cap, scores_dev,
                     (long long*)ids_dev, (const int*)nullptr);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int check_search_args(const amdr_dense* h, const void* Q, int nq, int k, const void* s, const void* i) {
  AMDR_REQUIRE(h != nullptr, "dense: null handle");
  AMDR_REQUIRE(nq >= 0, "dense: nq=%d", nq);
  AMDR_REQUIRE(k >= 1 && k <= AMDR_MAX_K, "dense: k=%d outside [1,%d]", k, AMDR_MAX_K);
  AMDR_REQUIRE(nq == 0 || (Q && s && i), "dense: null buffer");
  return AMDR_OK;
}

// Statistics of rows [row0, row0 + rows) folded into the handle's (max |component|, max row norm): what the fp16 first
// pass of large scans scales by and bounds its error with.  Synchronous (create / add are).
int update_stats(amdr_dense* h, int64_t row0, int64_t rows) {
  h->hi_ok = false;
  if (!dense_hi_supported(h->d)) return AMDR_OK;
  // max |x|, max row norm, then the counters: unresolved queries, flagged passes, passes, pad
  int rc = h->stats.ensure(8 * sizeof(unsigned int));
  if (rc) return rc;
  if (row0 == 0) AMDR_HIP(hipMemsetAsync(h->stats.p, 0, 8 * sizeof(unsigned int), h->stream));
  if (!h->hi_host) {
    AMDR_HIP(hipHostMalloc((void**)&h->hi_host, 4 * sizeof(unsigned int), hipHostMallocDefault));
    h->hi_host[0] = h->hi_host[1] = h->hi_host[2] = 0u;
    AMDR_HIP(hipEventCreateWithFlags(&h->hi_ev, hipEventDisableTiming));
  }
  if ((rc = dense_stats_launch(h->X + (size_t)row0 * h->d, (long)rows, h->d, h->stats.as<unsigned int>(), h->stream)))
    return rc;
  float st[2] = {0.f, 0.f};
  AMDR_HIP(hipMemcpyAsync(st, h->stats.p, sizeof(st), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipStreamSynchronize(h->stream));
  int e = 0;
  if (st[0] > 0.f && st[0] <= FLT_MAX) (void)frexpf(st[0], &e);
  h->x_scale = ldexpf(1.f, -e);
  h->row_norm_max = st[1];
  h->hi_ok = st[0] <= FLT_MAX && st[1] <= FLT_MAX && e > -100 && e < 100;
  return AMDR_OK;
}

}  // namespace

namespace amdr {
int dense_small_raw(amdr_dense_t* h, int nq, DenseRaw* out) {
  const long ld = ((long)h->n + 31) / 32 * 32;
  int rc = h->smat[0].ensure((size_t)nq * (size_t)ld * sizeof(float));
  if (rc) return rc;
  out->X = h->X;
  out->n = (long)h->n;
  out->d = h->d;
  out->S = h->smat[0].as<float>();
  out->ld = ld;
  return AMDR_OK;
}
std::mutex& dense_mutex(amdr_dense_t* h) { return h->mu; }
int dense_device_of(const amdr_dense_t* h) { return h->device; }
}  // namespace amdr

extern "C" {

int amdr_dense_create(const float* X_host, int64_t n, int32_t d, int32_t device, amdr_dense_t** out) {
  AMDR_REQUIRE(out != nullptr, "dense_create: out is null");
  *out = nullptr;
  AMDR_REQUIRE(n >= 0 && n < (1ll << 32), "dense_create: n=%lld outside [0, 2^32)", (long long)n);
  AMDR_REQUIRE(d >= 4 && d <= AMDR_MAX_DIM && d % 4 == 0, "dense_create: d=%d must be a multiple of 4 in [4,%d]", d,
               AMDR_MAX_DIM);
  AMDR_REQUIRE(n == 0 || X_host != nullptr, "dense_create: X is null");
  int rc = check_device(device);
  if (rc) return rc;
  amdr_dense* h = new (std::nothrow) amdr_dense();
  if (!h) return fail(AMDR_ENOMEM, "dense_create: host alloc");
  h->device = device;
  h->n = n;
  h->d = d;
  h->cap_rows = n;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess && n > 0) e = hipMalloc((void**)&h->X, (size_t)n * d * sizeof(float));
  if (e == hipSuccess && n > 0) e = hipMemcpy(h->X, X_host, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    amdr_dense_destroy(h);
    return fail(e == hipErrorOutOfMemory ? AMDR_ENOMEM : AMDR_EHIP, "dense_create: %s", hipGetErrorString(e));
  }
  if ((rc = update_stats(h, 0, n))) {
    amdr_dense_destroy(h);
    return rc;
  }
  *out = h;
  return AMDR_OK;
}

int amdr_dense_create_from_device(const float* X_dev, int64_t n, int32_t d, int32_t device, amdr_dense_t** out) {
  AMDR_REQUIRE(out != nullptr, "dense_create_from_device: out is null");
  *out = nullptr;
  AMDR_REQUIRE(n >= 0 && n < (1ll << 32), "dense_create_from_device: n=%lld outside [0, 2^32)", (long long)n);
  AMDR_REQUIRE(d >= 4 && d <= AMDR_MAX_DIM && d % 4 == 0, "dense_create_from_device: bad d=%d", d);
  AMDR_REQUIRE(n == 0 || X_dev != nullptr, "dense_create_from_device: X is null");
  AMDR_REQUIRE(((uintptr_t)X_dev & 15) == 0, "dense_create_from_device: X must be 16-byte aligned");
  int rc = check_device(device);
  if (rc) return rc;
  amdr_dense* h = new (std::nothrow) amdr_dense();
  if (!h) return fail(AMDR_ENOMEM, "dense_create_from_device: host alloc");
  h->device = device;
  h->n = n;
  h->d = d;
  h->cap_rows = n;
  h->X = const_cast<float*>(X_dev);
  h->owns = false;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    amdr_dense_destroy(h);
    return fail(AMDR_EHIP, "dense_create_from_device: %s", hipGetErrorString(e));
  }
  // The statistics kernel runs on the handle's own (non-blocking) stream: whatever produced X on ANOTHER stream must
  // have finished first, or the largest component / row norm — the fp16 first pass's error bound — would be taken from
  // a half-written matrix.  Creation is synchronous anyway: wait for the device.
  AMDR_HIP(hipDeviceSynchronize());
  if ((rc = update_stats(h, 0, n))) {  // the wrapped matrix must not change while the handle lives
    amdr_dense_destroy(h);
    return rc;
  }
  *out = h;
  return AMDR_OK;
}

int amdr_dense_add(amdr_dense_t* h, const float* X_host, int64_t n_add) {
  AMDR_REQUIRE(h != nullptr, "dense_add: null handle");
  AMDR_REQUIRE(h->owns, "dense_add: handle wraps caller-owned memory");
  AMDR_REQUIRE(n_add >= 0 && (n_add == 0 || X_host), "dense_add: bad arguments");
  AMDR_REQUIRE(h->n + n_add < (1ll << 32), "dense_add: too many rows");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  if (n_add == 0) return AMDR_OK;
  if (h->n + n_add > h->cap_rows) {
    int64_t ncap = h->cap_rows * 2 > h->n + n_add ? h->cap_rows * 2 : h->n + n_add;
    float* nx = nullptr;
    AMDR_HIP(hipMalloc((void**)&nx, (size_t)ncap * h->d * sizeof(float)));
    if (h->n > 0) AMDR_HIP(hipMemcpy(nx, h->X, (size_t)h->n * h->d * sizeof(float), hipMemcpyDeviceToDevice));
    if (h->X) (void)hipFree(h->X);
    h->X = nx;
    h->cap_rows = ncap;
  }
  AMDR_HIP(hipMemcpy(h->X + (size_t)h->n * h->d, X_host, (size_t)n_add * h->d * sizeof(float), hipMemcpyHostToDevice));
  const int64_t row0 = h->n;
  h->n += n_add;
  if (h->small) {  // the short-corpus fp16 image is of the old matrix (and may point at freed memory): made again on demand
    (void)hipDeviceSynchronize();
    (void)amdr_dense_small_destroy(h->small);
    h->small = nullptr;
  }
  h->small_failed = false;
  // the matrix changed: what the fp16 first pass learnt about it (width level, given up) starts over
  h->hi_level = 0;
  h->hi_off = false;
  h->lvl_p0 = h->hi_passes;
  if (h->hi_copy_pending && hipEventSynchronize(h->hi_ev) == hipSuccess) {
    h->hi_copy_pending = false;
    for (int i = 0; i < 3; ++i) h->hi_seen[i] = h->hi_host[i];
  }
  h->lvl_f0 = h->hi_seen[1];
  if (hi_tail2()) h->lvl_p0 = (int64_t)h->hi_seen[2];
  return update_stats(h, row0, n_add);
}

int amdr_dense_ntotal(const amdr_dense_t* h, int64_t* n) {
  AMDR_REQUIRE(h && n, "dense_ntotal: null");
  *n = h->n;
  return AMDR_OK;
}
int amdr_dense_dim(const amdr_dense_t* h, int32_t* d) {
  AMDR_REQUIRE(h && d, "dense_dim: null");
  *d = h->d;
  return AMDR_OK;
}

int amdr_dense_reserve(amdr_dense_t* h, int32_t nq_max, int32_t k_max) {
  AMDR_REQUIRE(h != nullptr, "dense_reserve: null handle");
  AMDR_REQUIRE(nq_max >= 1 && k_max >= 1 && k_max <= AMDR_MAX_K, "dense_reserve: bad sizes");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  ScanPlan p;
  make_plan(h->n, h->d, nq_max, k_max, &p);
  int rc = h->part[0].ensure(p.part_bytes);
  if (rc) return rc;
  // batches on a large matrix take the two-level form (every batch size <= nq_max and depth <= k_max that does is
  // covered, see two_level_ensure); longer ones the panel kernel + score matrix
  if ((rc = two_level_ensure(h, 0, nq_max, k_max, true))) return rc;
  if (hi_applies(h, nq_max, k_max) || two_level_applies(h, nq_max, k_max)) {
  } else if (nq_max >= kBatchedMin && h->n > 0 && dense_mfma_supported(h->d)) {
    DenseMfmaPlan mp;
    const int cmax = batched_chunk(h, nq_max);
    dense_mfma_plan((long)h->n, h->d, cmax, k_max, &mp);
    if ((rc = h->smat[0].ensure(mp.s_bytes))) return rc;
    size_t part_need = 0;  // every pass size a call within nq_max can have (slab lists: not monotone in the size)
    for (int m = 1; m <= cmax; ++m) {
      dense_mfma_plan((long)h->n, h->d, m, k_max, &mp);
      part_need = mp.part_bytes > part_need ? mp.part_bytes : part_need;
    }
    if ((rc = h->part[0].ensure(part_need))) return rc;
  }
  if (h->n > 0 && h->n <= kRowWavesMax) {  // the 1-4 query call on a short corpus also goes through S
    DenseMfmaPlan mp;
    dense_mfma_plan((long)h->n, h->d, nq_max < kBatchedMin ? nq_max : kBatchedMin - 1, k_max, &mp);
    if ((rc = h->smat[0].ensure(mp.s_bytes))) return rc;
    if ((rc = h->part[0].ensure(mp.part_bytes))) return rc;
  }
  if ((rc = h->qbuf.ensure((size_t)nq_max * h->d * sizeof(float)))) return rc;
  if ((rc = h->sbuf.ensure((size_t)nq_max * k_max * sizeof(float)))) return rc;
  if (small_hi_shape(h, nq_max, k_max < 12 ? k_max : 12, 0))  // (so that a later capture finds the two-pass form's buffers)
    (void)small_hi_ready(h, batched_chunk(h, nq_max) < nq_max ? batched_chunk(h, nq_max) : nq_max, nullptr);
  return h->ibuf.ensure((size_t)nq_max * k_max * sizeof(int64_t));
}

int amdr_dense_search_device(amdr_dense_t* h, const float* Q_dev, int32_t nq, int32_t k, float* scores_dev,
                             int64_t* ids_dev, void* stream) {
  int rc = check_search_args(h, Q_dev, nq, k, scores_dev, ids_dev);
  if (rc) return rc;
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  return run_search(h, 0, Q_dev, nq, k, scores_dev, ids_dev, (hipStream_t)stream);
}

int amdr_dense_search_fuse_device(amdr_dense_t* h, const float* Q_dev, int32_t nq, int32_t k,
                                  const amdr_fuse_params_t* p, const int64_t* dense_row2uid, const int64_t* bm25_ids,
                                  const double* bm25_scores, int32_t kb, const int64_t* bm25_row2uid,
                                  float* dense_scores_dev, int64_t* dense_ids_dev, int64_t* out_ids, double* out_vals,
                                  int32_t* out_mask, int32_t* out_count, void* stream) {
  int rc = check_search_args(h, Q_dev, nq, k, dense_scores_dev, dense_ids_dev);
  if (rc) return rc;
  AMDR_REQUIRE(p != nullptr, "dense_search_fuse: null params");
  AMDR_REQUIRE(p->method >= 0 && p->method <= AMDR_FUSE_WEIGHTED_SUM, "dense_search_fuse: method=%d", p->method);
  AMDR_REQUIRE(kb >= 0 && kb <= AMDR_MAX_K && (kb == 0 || (bm25_ids && bm25_scores)), "dense_search_fuse: bad BM25 lists");
  AMDR_REQUIRE(nq == 0 || (out_ids && out_vals && out_mask && out_count), "dense_search_fuse: null output");
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  FuseTail tail{p, dense_row2uid, bm25_ids, bm25_scores, kb, bm25_row2uid, out_ids, out_vals, out_mask, out_count};
  if (!hi_applies(h, nq, k) && !two_level_applies(h, nq, k) && nq >= kBatchedMin && h->n > 0 &&
      dense_mfma_supported(h->d))
    return run_search_batched(h, 0, Q_dev, nq, k, dense_scores_dev, dense_ids_dev, st, false, &tail);
  if ((rc = run_search(h, 0, Q_dev, nq, k, dense_scores_dev, dense_ids_dev, st))) return rc;
  return dense_fuse_plain_launch(tail, 0, nq, k, dense_scores_dev, dense_ids_dev, st);
}

int amdr_dense_search(amdr_dense_t* h, const float* Q_host, int32_t nq, int32_t k, float* scores_host,
                      int64_t* ids_host) {
  int rc = check_search_args(h, Q_host, nq, k, scores_host, ids_host);
  if (rc) return rc;
  if (nq == 0) return AMDR_OK;
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  if ((rc = h->qbuf.ensure((size_t)nq * h->d * sizeof(float)))) return rc;
  if ((rc = h->sbuf.ensure((size_t)nq * k * sizeof(float)))) return rc;
  if ((rc = h->ibuf.ensure((size_t)nq * k * sizeof(int64_t)))) return rc;
  AMDR_HIP(hipMemcpyAsync(h->qbuf.p, Q_host, (size_t)nq * h->d * sizeof(float), hipMemcpyHostToDevice, h->stream));
  rc = run_search(h, 1, h->qbuf.as<float>(), nq, k, h->sbuf.as<float>(), h->ibuf.as<int64_t>(), h->stream);
  if (rc) return rc;
  AMDR_HIP(hipMemcpyAsync(scores_host, h->sbuf.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipMemcpyAsync(ids_host, h->ibuf.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipStreamSynchronize(h->stream));
  return AMDR_OK;
}

int amdr_dense_score_rows(amdr_dense_t* h, const float* Q_host, int32_t nq, const int64_t* rows_host, int32_t m,
                          float* scores_host) {
  AMDR_REQUIRE(h != nullptr, "dense_score_rows: null handle");
  AMDR_REQUIRE(nq >= 0 && m >= 0, "dense_score_rows: bad sizes");
  if (nq == 0 || m == 0) return AMDR_OK;
  AMDR_REQUIRE(Q_host && rows_host && scores_host, "dense_score_rows: null buffer");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  int rc;
  const size_t cnt = (size_t)nq * m;
  if ((rc = h->qbuf.ensure((size_t)nq * h->d * sizeof(float)))) return rc;
  if ((rc = h->sbuf.ensure(cnt * sizeof(float)))) return rc;
  if ((rc = h->ibuf.ensure(cnt * sizeof(int64_t)))) return rc;
  AMDR_HIP(hipMemcpyAsync(h->qbuf.p, Q_host, (size_t)nq * h->d * sizeof(float), hipMemcpyHostToDevice, h->stream));
  AMDR_HIP(hipMemcpyAsync(h->ibuf.p, rows_host, cnt * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(dense_score_rows_kernel, dim3(ceil_div((long)cnt, kWaves)), dim3(256), 0, h->stream, h->X,
                     (long)h->n, h->d, h->qbuf.as<float>(), nq, h->ibuf.as<long long>(), m, h->sbuf.as<float>());
  AMDR_HIP(hipGetLastError());
  AMDR_HIP(hipMemcpyAsync(scores_host, h->sbuf.p, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  AMDR_HIP(hipStreamSynchronize(h->stream));
  return AMDR_OK;
}

int amdr_dense_read_rows(const amdr_dense_t* h, int64_t row0, int64_t nrows, float* out_host) {
  AMDR_REQUIRE(h && out_host, "dense_read_rows: null");
  AMDR_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= h->n, "dense_read_rows: range outside [0,%lld)",
               (long long)h->n);
  AMDR_HIP(hipSetDevice(h->device));
  if (nrows)
    AMDR_HIP(hipMemcpy(out_host, h->X + (size_t)row0 * h->d, (size_t)nrows * h->d * sizeof(float),
                       hipMemcpyDeviceToHost));
  return AMDR_OK;
}

int amdr_dense_workspace_plan(int64_t n, int32_t d, int32_t nq, int32_t k, int64_t* out6) {
  AMDR_REQUIRE(out6 != nullptr, "dense_workspace_plan: null");
  AMDR_REQUIRE(n >= 1 && n < (1ll << 32) && d >= 4 && d <= AMDR_MAX_DIM && d % 4 == 0, "dense_workspace_plan: bad shape");
  AMDR_REQUIRE(nq >= 1 && k >= 1 && k <= AMDR_MAX_K, "dense_workspace_plan: bad sizes");
  amdr_dense h;  // shape only: no device is touched
  h.n = n;
  h.d = d;
  for (int i = 0; i < 6; ++i) out6[i] = 0;
  auto mx = [](int64_t& a, size_t b) { a = (int64_t)b > a ? (int64_t)b : a; };
  h.hi_ok = dense_hi_supported(d);  // shape only: the statistics of a real matrix can only take the fp16 first pass away
  if (hi_applies(&h, nq, k) || two_level_applies(&h, nq, k)) {
    TwoLevelNeed need, used;
    two_level_need(&h, nq, k, &need);
    out6[0] = (int64_t)need.smat, out6[1] = (int64_t)need.part, out6[2] = (int64_t)need.aux;
    const bool hi = hi_applies(&h, nq, k);
    if (hi && hi_tail2()) {  // the round-4 tail: passes of up to four query tiles, any of the three widths
      const int c2 = hi2_chunk(&h, nq);
      for (int q0 = 0; q0 < nq; q0 += c2)
        for (int l = 0; l < kHiLevels; ++l) {
          Hi2Plan p;
          hi2_plan(&h, nq - q0 < c2 ? nq - q0 : c2, k, hi_kc(k, l), &p);
          used.smat = p.smat_bytes > used.smat ? p.smat_bytes : used.smat;
          used.aux = p.aux_bytes + 256 > used.aux ? p.aux_bytes + 256 : used.aux;
        }
    }
    const int chunk = hi ? hi_chunk(&h, nq, k) : two_level_chunk(nq, k);
    for (int q0 = 0; q0 < nq; q0 += chunk) {  // what the pass loop of run_search_two_level touches (round-3 tail / given up)
      const int m = nq - q0 < chunk ? nq - q0 : chunk;
      TwoLevelPlan t;
      if (hi)
        for (int l = 0; l < kHiLevels; ++l) {
          two_level_plan(&h, m, k, hi_kc(k, l), &t);
          used.add(t);
        }
      const int ec = hi ? two_level_chunk(m, k) : m;
      for (int e0 = 0; e0 < m; e0 += ec) {
        two_level_plan(&h, m - e0 < ec ? m - e0 : ec, k, k, &t);
        used.add(t);
      }
    }
    out6[3] = (int64_t)used.smat, out6[4] = (int64_t)used.part, out6[5] = (int64_t)used.aux;
  } else if (nq >= kBatchedMin && dense_mfma_supported(d)) {
    const int chunk = batched_chunk(&h, nq);
    DenseMfmaPlan p;
    dense_mfma_plan((long)n, d, chunk, k, &p);
    out6[0] = (int64_t)p.s_bytes, out6[1] = (int64_t)batched_part_need(&h, nq, k);
    for (int q0 = 0; q0 < nq; q0 += chunk) {
      dense_mfma_plan((long)n, d, nq - q0 < chunk ? nq - q0 : chunk, k, &p);
      mx(out6[3], p.s_bytes), mx(out6[4], p.part_bytes);
    }
  }
  return AMDR_OK;
}

int amdr_dense_plan_info(const amdr_dense_t* h, int32_t nq, int32_t k, char* buf, int32_t buf_len) {
  AMDR_REQUIRE(h && buf && buf_len > 0, "dense_plan_info: null");
  AMDR_REQUIRE(nq >= 1 && k >= 1 && k <= AMDR_MAX_K, "dense_plan_info: bad sizes");
  if (h->n <= 0) {
    snprintf(buf, buf_len, "empty index");
    return AMDR_OK;
  }
  if (hi_applies(h, nq, k) && !h->hi_off && hi_tail2()) {
    const int m = hi2_chunk(h, nq), kc = hi_kc(k, hi_level_of(h));
    Hi2Plan p;
    hi2_plan(h, m, k, kc, &p);
    snprintf(buf, buf_len,
             "dense_hi_tilemax_kernel fp16 first pass queries_per_launch=%d scans_per_launch=%d (%d per scan, one tail): per-query lists of the "
             "approximate tile maxima above a sampled threshold (every %ld-th tile, width level %d) -> top-%d + rounding-bound "
             "check + exact re-scoring of each query's tiles at or above its cut (<= %d each) + top-k: 4 launches behind the "
             "scan(s), the exact first pass behind a device flag in 2",
             m, p.qtiles, m < dense_hi_max_queries(h->d) ? m : dense_hi_max_queries(h->d),
             dense_hi2_sample_stride((long)h->n, p.qtiles), hi_level_of(h), kc, kc);
    return AMDR_OK;
  }
  if (hi_applies(h, nq, k) && !h->hi_off) {
    TwoLevelPlan t;
    const int m = hi_chunk(h, nq, k), kc = hi_kc(k, hi_level_of(h));
    two_level_plan(h, m, k, kc, &t);
    snprintf(buf, buf_len,
             "dense_hi_tilemax_kernel fp16 first pass queries_per_launch=%d two-level: top-%d of %ld approximate tile "
             "maxima (width level %d; threshold from a sample of every %ld-th tile), cut checked against the rounding bound "
             "+ exact re-scoring of every query's tiles at or above its cut (<= %d each) + top-k (exact first pass behind a "
             "device flag)",
             m, kc, t.tiles, hi_level_of(h), dense_hi_sample_stride((long)h->n), kc);
    return AMDR_OK;
  }
  if (hi_applies(h, nq, k)) {  // the handle gave the fp16 pass up: exact passes in the same shapes
    TwoLevelPlan t;
    const int m = two_level_chunk(hi_chunk(h, nq, k), k);
    two_level_plan(h, m, k, k, &t);
    snprintf(buf, buf_len,
             "dense_mfma_scores_kernel tile-maxima grid=%dx%d queries_per_launch=%d two-level: top-%d of %ld tile maxima "
             "+ re-scoring of <= %d candidate tiles + top-k (fp16 first pass given up: too many unresolved queries)",
             t.scan.grid_x, t.scan.grid_y, m, k, t.tiles, m * k);
    return AMDR_OK;
  }
  if (two_level_applies(h, nq, k)) {
    TwoLevelPlan t;
    const int m = two_level_chunk(nq, k);
    two_level_plan(h, m, k, k, &t);
    snprintf(buf, buf_len,
             "dense_mfma_scores_kernel tile-maxima grid=%dx%d queries_per_launch=%d two-level: top-%d of %ld tile maxima "
             "+ re-scoring of <= %d candidate tiles + top-k",
             t.scan.grid_x, t.scan.grid_y, m, k, t.tiles, m * k);
    return AMDR_OK;
  }
  const bool batched = nq >= kBatchedMin && dense_mfma_supported(h->d);
  if (batched || h->n <= kRowWavesMax) {
    const int m = batched_chunk(h, nq);
    DenseMfmaPlan p;
    dense_mfma_plan((long)h->n, h->d, m, k, &p);
    if (batched && p.slabs == 1 && small_hi_shape(h, m < nq ? m : nq, k, 0) && !h->small_failed) {
      snprintf(buf, buf_len,
               "dsh_scores_kernel fp16 first pass queries_per_launch=%d (dsh_split_queries_kernel + v_mfma_f32_32x32x16_f16 on "
               "fp16 roundings of both operands, proven per-query bound) + dense_hi_select_fuse_kernel (rows inside 2 eps of the "
               "k-th best re-scored exactly, top-k, fusion); exact form: dense_panel_scores_kernel",
               m < nq ? m : nq);
      return AMDR_OK;
    }
    const char* tail = p.slabs == 1 ? "scores_slab_topk_kernel" : "scores_slab_topk_kernel + dense_merge_kernel";
    if (!batched) {
      snprintf(buf, buf_len, "dense_all_scores_kernel (one wave per query x row) + %s", tail);
    } else if (dense_panel_supported((long)h->n, h->d, m)) {
      DensePanelPlan pp;
      dense_panel_plan((long)h->n, h->d, m, &pp);
      snprintf(buf, buf_len, "dense_panel_scores_kernel nb=%d parts=%d blocks=%d queries_per_launch=%d + %s", pp.nb,
               pp.parts, pp.m_tiles * pp.parts, m, tail);
    } else {
      snprintf(buf, buf_len, "dense_mfma_scores_kernel query-tiles-in-LDS grid=%dx%d queries_per_launch=%d + %s",
               p.grid_x, p.grid_y, m, tail);
    }
    return AMDR_OK;
  }
  ScanPlan p;
  make_plan(h->n, h->d, nq, k, &p);
  snprintf(buf, buf_len, "dense_scan_topk_kernel<NQ=%d> grid=%dx%d%s", p.nq_per_block, p.grid_x, p.grid_y,
           p.grid_x == 1 ? "" : " + dense_merge_kernel");
  return AMDR_OK;
}

int amdr_dense_hi_counters(amdr_dense_t* h, int64_t* out6) {
  AMDR_REQUIRE(h && out6, "dense_hi_counters: null");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  out6[0] = h->hi_queries;
  out6[1] = out6[5] = 0;
  if (h->stats.p) {
    unsigned int c[3] = {0, 0, 0};
    AMDR_HIP(hipDeviceSynchronize());  // the counters are bumped by kernels on the callers' streams
    AMDR_HIP(hipMemcpy(c, h->stats.as<unsigned int>() + 2, sizeof(c), hipMemcpyDeviceToHost));
    out6[1] = (int64_t)c[0];
    out6[5] = (int64_t)c[1];
  }
  out6[2] = hi_level_of(h);
  out6[3] = h->hi_ok && !h->hi_off ? 1 : 0;
  out6[4] = h->hi_passes;
  return AMDR_OK;
}

int amdr_dense_profile_begin(amdr_dense_t* h, int32_t max_launches) {
  AMDR_REQUIRE(h != nullptr && max_launches >= 1 && max_launches <= (1 << 16), "dense_profile_begin: bad arguments");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  while (h->prof_ev.size() < (size_t)max_launches * 2) {
    hipEvent_t e;
    AMDR_HIP(hipEventCreate(&e));
    h->prof_ev.push_back(e);
  }
  h->prof_used = 0;
  h->prof_on = true;
  return AMDR_OK;
}

int amdr_dense_profile_end(amdr_dense_t* h, double* total_ms, int32_t* launches) {
  AMDR_REQUIRE(h && total_ms && launches, "dense_profile_end: null");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  h->prof_on = false;
  double tot = 0;
  for (int i = 0; i + 1 < h->prof_used; i += 2) {
    float ms = 0;
    AMDR_HIP(hipEventSynchronize(h->prof_ev[i + 1]));
    AMDR_HIP(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = h->prof_used / 2;
  h->prof_used = 0;
  return AMDR_OK;
}

int amdr_dense_two_pass_fallbacks(amdr_dense_t* h, int64_t* out) {
  AMDR_REQUIRE(h && out, "dense_two_pass_fallbacks: null");
  *out = 0;
  std::lock_guard<std::mutex> g(h->mu);
  if (!h->small_fb.p) return AMDR_OK;
  AMDR_HIP(hipSetDevice(h->device));
  unsigned int v = 0;
  AMDR_HIP(hipMemcpy(&v, h->small_fb.p, sizeof(v), hipMemcpyDeviceToHost));  // (synchronises with the device)
  *out = (int64_t)v;
  return AMDR_OK;
}

int amdr_dense_destroy(amdr_dense_t* h) {
  if (!h) return AMDR_OK;
  (void)hipSetDevice(h->device);
  for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
  if (h->stream) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamDestroy(h->stream);
  }
  if (h->owns && h->X) (void)hipFree(h->X);
  for (int w = 0; w < 2; ++w) {
    h->part[w].release();
    h->smat[w].release();
    h->aux[w].release();
  }
  h->qbuf.release();
  h->sbuf.release();
  h->ibuf.release();
  h->stats.release();
  if (h->small) (void)amdr_dense_small_destroy(h->small);
  h->small_eps.release();
  h->small_fb.release();
  if (h->hi_host) (void)hipHostFree(h->hi_host);
  if (h->hi_ev) (void)hipEventDestroy(h->hi_ev);
  delete h;
  return AMDR_OK;
}

int amdr_merge_topk_f32_device(const float* scores, const int64_t* ids, int32_t n_parts, int32_t nq, int32_t k_in,
                               int32_t k_out, float* out_scores, int64_t* out_ids, int32_t device, void* stream) {
  AMDR_REQUIRE(scores && ids && out_scores && out_ids, "merge_topk: null buffer");
  AMDR_REQUIRE(n_parts >= 1 && nq >= 1 && k_in >= 1 && k_out >= 1 && k_out <= AMDR_MAX_K, "merge_topk: bad sizes");
  AMDR_HIP(hipSetDevice(device));
  return launch_merge_parts<float>(scores, ids, n_parts, nq, k_in, k_out, out_scores, out_ids, (hipStream_t)stream);
}
int amdr_merge_topk_f64_device(const double* scores, const int64_t* ids, int32_t n_parts, int32_t nq, int32_t k_in,
                               int32_t k_out, double* out_scores, int64_t* out_ids, int32_t device, void* stream) {
  AMDR_REQUIRE(scores && ids && out_scores && out_ids, "merge_topk: null buffer");
  AMDR_REQUIRE(n_parts >= 1 && nq >= 1 && k_in >= 1 && k_out >= 1 && k_out <= AMDR_MAX_K, "merge_topk: bad sizes");
  AMDR_HIP(hipSetDevice(device));
  return launch_merge_parts<double>(scores, ids, n_parts, nq, k_in, k_out, out_scores, out_ids, (hipStream_t)stream);
}

}  // extern "C"
