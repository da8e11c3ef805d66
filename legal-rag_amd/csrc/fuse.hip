// Fusion + min_final filter + rerank blend, on device, fp64, no FMA contraction.
//
// Replaces HybridRetriever._fuse (legalrag/retrieval/hybrid_retriever.py:389-551),
// _minmax (:24-30), _rrf_with_breakdown (:33-56), the min_final_score filter
// (:309-310) and the rerank blend (:338-355, rerankers.py:48-54,349).
// One wave per query; the candidate set is at most kd+kb+kc (<= 768) ids, so
// this is latency work: the point of doing it on the GPU is that a batch of
// queries never leaves HBM between the channel kernels and the final top-k.
// Every expression below is written in the reference's operand order and this
// file is compiled with -ffp-contract=off so results are bit-identical to the
// Python float arithmetic.  Exactly tied scores keep first-appearance order.
#include "bm25_core.hpp"
#include "common.hpp"
#include "dense_dot.hpp"
#include "topk.hpp"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace amdr {

constexpr int kFuseMax = 3 * AMDR_MAX_K;  // 768

// all-lanes reductions on DPP / permlane-swap exchanges (topk.hpp), not ds_bpermute
__device__ __forceinline__ double wave_min(double v) { return wave_allmin_f64(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_allmax_f64(v); }
__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct ChanIn {
  const long long* ids;  // [nq, k]
  const void* scores;    // float or double [nq, k]
  const long long* row2uid;
  int k;
  int is_f64;
};

__device__ __forceinline__ double chan_score(const ChanIn& c, int qi, int j) {
  size_t off = (size_t)qi * c.k + j;
  return c.is_f64 ? ((const double*)c.scores)[off] : (double)((const float*)c.scores)[off];
}
__device__ __forceinline__ long long chan_uid(const ChanIn& c, int qi, int j) {
  long long id = c.ids[(size_t)qi * c.k + j];
  if (id >= 0 && c.row2uid) id = c.row2uid[id];
  return id;
}

// Everything _fuse reports for one candidate (hybrid_retriever.py:389-551), operand for operand
// in the reference's order; shared by the one-query-per-wave kernel and the packed one.
struct FuseCtx {
  double rmn, rmx;  // min / max of the RRF totals over the union
  bool rdeg, wrrf;
  double w[3], lo[3], hi[3];  // channel weight, min and max of the channel's scores
};
template <class ScoreAt>
__device__ __forceinline__ void fuse_eval(const amdr_fuse_params_t& P, const FuseCtx& X, double t, const int (&pos)[3],
                                          ScoreAt&& score_at, double (&val)[AMDR_FUSE_NVALS], int& mk) {
  const double rrf_norm = X.rdeg ? 0.0 : (t - X.rmn) / (X.rmx - X.rmn);
  double nrm[3], raw[3], wt[3];
  mk = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int p = pos[c];
    nrm[c] = 0.0;
    raw[c] = 0.0;
    if (p >= 0) {
      mk |= (1 << c);
      const double s = score_at(c, p);
      nrm[c] = (X.hi[c] - X.lo[c] < 1e-12) ? 0.0 : (s - X.lo[c]) / (X.hi[c] - X.lo[c]);
      const double wc = X.wrrf ? X.w[c] : 1.0;
      raw[c] = wc * (1.0 / (double)(P.rrf_k + p + 1));
    }
    wt[c] = X.w[c] * nrm[c];
  }
  const double wsum = (wt[0] + wt[1]) + wt[2];
  double score, con[3] = {0.0, 0.0, 0.0};
  if (P.method == AMDR_FUSE_WEIGHTED_SUM) {
    score = wsum;
#pragma unroll
    for (int c = 0; c < 3; ++c) con[c] = wt[c];
  } else if (P.method == AMDR_FUSE_RRF || P.method == AMDR_FUSE_WRRF) {
    score = rrf_norm;
    const double mass = score;
    if (!(mass <= 0.0 || t <= 1e-18)) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (pos[c] >= 0) con[c] = mass * raw[c] / t;
    }
  } else {
    score = P.alpha * rrf_norm + (1.0 - P.alpha) * wsum;
#pragma unroll
    for (int c = 0; c < 3; ++c) con[c] = 0.0 + (1.0 - P.alpha) * wt[c];
    const double mass = P.alpha * rrf_norm;
    if (!(mass <= 0.0 || t <= 1e-18)) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (pos[c] >= 0) con[c] = con[c] + mass * raw[c] / t;
    }
  }
  val[AMDR_FV_SCORE] = score;
  val[AMDR_FV_RRF_NORM] = rrf_norm;
  val[AMDR_FV_WSUM] = wsum;
  val[AMDR_FV_NORM_DENSE] = nrm[0];
  val[AMDR_FV_NORM_BM25] = nrm[1];
  val[AMDR_FV_NORM_COLBERT] = nrm[2];
  val[AMDR_FV_CONTRIB_DENSE] = con[0];
  val[AMDR_FV_CONTRIB_BM25] = con[1];
  val[AMDR_FV_CONTRIB_COLBERT] = con[2];
}

// block = 64 threads (one wave); grid = nq
__global__ __launch_bounds__(64) void fuse_kernel(amdr_fuse_params_t P, ChanIn c0, ChanIn c1, ChanIn c2, int max_out,
                                                  long long* __restrict__ out_ids, double* __restrict__ out_vals,
                                                  int* __restrict__ out_mask, int* __restrict__ out_count) {
  // dynamic LDS sized by max_out (<= kFuseMax): 36 bytes per candidate
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  long long* uid = reinterpret_cast<long long*>(fsm);
  double* sc = reinterpret_cast<double*>(uid + max_out);
  double* tot = sc + max_out;
  int* pos0 = reinterpret_cast<int*>(tot + max_out);
  int* pos[3] = {pos0, pos0 + max_out, pos0 + 2 * max_out};
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  const ChanIn ch[3] = {c0, c1, c2};
  const double w[3] = {P.w_dense, P.w_bm25, P.w_colbert};

  // ---- valid prefix length, min / max of every channel ------------------
  int n[3];
  double lo[3], hi[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int cnt = 0;
    double mn = INFINITY, mx = -INFINITY;
    for (int j = lane; j < ch[c].k; j += 64) {
      if (ch[c].ids[(size_t)qi * ch[c].k + j] >= 0) {
        cnt++;
        double s = chan_score(ch[c], qi, j);
        mn = fmin(mn, s);
        mx = fmax(mx, s);
      }
    }
    n[c] = wave_allsum_i32(cnt);
    lo[c] = wave_min(mn);
    hi[c] = wave_max(mx);
  }

  // ---- union of ids in first-appearance order ----------------------------
  int U = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int U0 = U;  // entries a new id can collide with (ids are unique inside a channel)
    for (int base = 0; base < n[c]; base += 64) {
      const int j = base + lane;
      const bool v = j < n[c];
      long long my = v ? chan_uid(ch[c], qi, j) : -1;
      int f = -1;
      if (v)
        for (int u = 0; u < U0; ++u)
          if (uid[u] == my) {
            f = u;
            break;
          }
      const bool isnew = v && f < 0;
      const unsigned long long m = __ballot(isnew);
      const unsigned long long lt = (lane == 0) ? 0ull : (m & (~0ull >> (64 - lane)));
      const int idx = isnew ? U + __popcll(lt) : f;
      if (isnew) {
        uid[idx] = my;
        pos[0][idx] = -1;
        pos[1][idx] = -1;
        pos[2][idx] = -1;
      }
      lds_sync();
      if (v) pos[c][idx] = j;
      U += __popcll(m);
      lds_sync();
    }
  }

  // ---- RRF totals, their min / max ---------------------------------------
  const bool wrrf = (P.method == AMDR_FUSE_WRRF);
  double rmn = INFINITY, rmx = -INFINITY;
  for (int u = lane; u < U; u += 64) {
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int p = pos[c][u];
      if (p >= 0) {
        double wc = wrrf ? w[c] : 1.0;
        double v = wc * (1.0 / (double)(P.rrf_k + p + 1));
        t = t + v;
      }
    }
    tot[u] = t;
    rmn = fmin(rmn, t);
    rmx = fmax(rmx, t);
  }
  rmn = wave_min(rmn);
  rmx = wave_max(rmx);
  const bool rdeg = (rmx - rmn < 1e-12);

  // all values of candidate u (fuse_eval); the long-list path evaluates twice (score pass,
  // output pass) so that nothing but the score has to live across the rank computation
  FuseCtx X;
  X.rmn = rmn;
  X.rmx = rmx;
  X.rdeg = rdeg;
  X.wrrf = wrrf;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    X.w[c] = w[c];
    X.lo[c] = lo[c];
    X.hi[c] = hi[c];
  }
  auto eval = [&](int u, double (&val)[AMDR_FUSE_NVALS], int& mk) {
    const int pp[3] = {pos[0][u], pos[1][u], pos[2][u]};
    fuse_eval(P, X, tot[u], pp, [&](int c, int p) { return chan_score(ch[c], qi, p); }, val, mk);
  };

  // ---- score, stable descending rank, filter, scatter ---------------------
  int kept = 0;
  const size_t obase = (size_t)qi * max_out;
  if (U <= 64) {
    // one candidate per lane (the serving shape: <= 3 x top-k ids): evaluate once, keep the nine
    // values in registers across the rank computation
    double val[AMDR_FUSE_NVALS];
    int mk = 0;
    const int u = lane;
    if (u < U) {
      eval(u, val, mk);
      sc[u] = val[AMDR_FV_SCORE];
    }
    lds_sync();
    if (u < U) {
      const double s = val[AMDR_FV_SCORE];
      int r = 0;
      for (int v2 = 0; v2 < U; ++v2) {
        const double o = sc[v2];
        r += (o > s) || (o == s && v2 < u);
      }
      if (s >= P.min_final_score) kept++;
      out_ids[obase + r] = uid[u];
      out_mask[obase + r] = mk;
#pragma unroll
      for (int x = 0; x < AMDR_FUSE_NVALS; ++x) out_vals[(obase + r) * AMDR_FUSE_NVALS + x] = val[x];
    }
  } else {
    for (int u = lane; u < U; u += 64) {
      double val[AMDR_FUSE_NVALS];
      int mk;
      eval(u, val, mk);
      sc[u] = val[AMDR_FV_SCORE];
    }
    lds_sync();
    for (int u = lane; u < U; u += 64) {
      double val[AMDR_FUSE_NVALS];
      int mk;
      eval(u, val, mk);
      const double s = val[AMDR_FV_SCORE];
      int r = 0;
      for (int v2 = 0; v2 < U; ++v2) {
        const double o = sc[v2];
        r += (o > s) || (o == s && v2 < u);
      }
      if (s >= P.min_final_score) kept++;
      out_ids[obase + r] = uid[u];
      out_mask[obase + r] = mk;
#pragma unroll
      for (int x = 0; x < AMDR_FUSE_NVALS; ++x) out_vals[(obase + r) * AMDR_FUSE_NVALS + x] = val[x];
    }
  }
  kept = wave_allsum_i32(kept);
  for (int r = U + lane; r < max_out; r += 64) {
    out_ids[obase + r] = -1;
    out_mask[obase + r] = 0;
    for (int x = 0; x < AMDR_FUSE_NVALS; ++x) out_vals[(obase + r) * AMDR_FUSE_NVALS + x] = 0.0;
  }
  if (lane == 0) out_count[qi] = kept;
}


// Packed form for the serving shape: when all candidates of a query fit in W lanes (max_out
// <= W; top-10 of two or three channels -> W = 32), a wave fuses 64 / W queries side by side,
// one candidate per lane.  Same arithmetic as fuse_kernel (fuse_eval), same outputs; the
// reductions run inside the W-lane group and the ballots are cut to the group's bits.  The
// kernel is bound by vector instructions issued per wave (fp64 divisions), not by data, so
// halving the waves halves its time.
// Entries a caller already holds in registers (lane sl = list position sl of its query): the fused dense top-k +
// fusion kernel hands over the dense list straight from its selector and the BM25 list it requested up front.
struct FusePre {
  bool have[3];
  long long id[3];  // raw channel id (-1 = padding), before row2uid
  double s[3];
};
template <int W, bool PRE>
__device__ __forceinline__ void fuse_packed_body(const amdr_fuse_params_t& P, const ChanIn& c0, const ChanIn& c1,
                                                 const ChanIn& c2, int nq, int max_out, long long* __restrict__ out_ids,
                                                 double* __restrict__ out_vals, int* __restrict__ out_mask,
                                                 int* __restrict__ out_count, const FusePre& pre, int qbase) {
  constexpr int G = 64 / W;  // queries per wave
  __shared__ long long s_uid[G][W];
  __shared__ double s_sc[G][W];
  __shared__ double s_chs[G][3][W];  // channel scores by list position
  __shared__ int s_pos[G][3][W];
  const int lane = threadIdx.x, seg = lane / W, sl = lane % W;
  const int qi = qbase + seg;  // (the packed kernels: blockIdx.x * G)
  const bool live = qi < nq;
  const ChanIn ch[3] = {c0, c1, c2};
  const double w[3] = {P.w_dense, P.w_bm25, P.w_colbert};
  const unsigned long long seg_bits = (W == 64) ? ~0ull : (((1ull << (W & 63)) - 1ull) << (seg * W));
  long long* uid = s_uid[seg];
  double* sc = s_sc[seg];

  // ---- per channel: valid prefix length, min / max, ids and scores -----------------------
  // A switched-off channel (k = 0, a kernel argument) is skipped as a whole; the counts are ballots, not lane
  // sums; and min / max of a channel whose scores arrive in descending order — the contract of
  // include/amdretrieval.h, and what _fuse's stable sort gives — are its first and last valid entries (one
  // neighbour compare + two lane reads instead of two five-step fp64 reductions; a list that is not descending,
  // NaNs included, still takes the reductions).  SQ counters: 857 -> 640 vector instructions per wave (42 % of
  // them were the cross-lane moves of the 64-bit reductions) — for 31.5 -> 30.7 us only: at 857 the kernel was
  // bound by vector issue, at 640 by the lifetime of its waves (two memory round trips, LDS exchanges, stores).
  // Not kept: two or more passes per wave with the next pass's lists prefetched (33.8 / 35.0 us).
  int n[3] = {0, 0, 0};
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  long long my_uid[3] = {-1, -1, -1};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (ch[c].k == 0) continue;
    const int j = sl;
    const bool inr = live && j < ch[c].k;
    long long id = -1;
    double s = 0.0;
    if (PRE && pre.have[c]) {
      if (inr) {
        id = pre.id[c];
        s = pre.s[c];
      }
    } else if (inr) {  // id and score are requested together: one memory round trip, not two
      id = ch[c].ids[(size_t)qi * ch[c].k + j];
      s = chan_score(ch[c], qi, j);
    }
    const bool has = inr && id >= 0;
    if (!has) s = 0.0;
    if (has && ch[c].row2uid) id = ch[c].row2uid[id];
    my_uid[c] = has ? id : -1;
    s_chs[seg][c][sl] = s;
    const unsigned long long hm = __ballot(has) & seg_bits;
    n[c] = __popcll(hm);
    // descending prefix?  valid entries form a prefix; lane sl compares with its right neighbour
    const double nxt = __shfl_down(s, 1);
    const bool in_order = !(has && sl + 1 < n[c]) || s >= nxt;
    const bool prefix = hm == (seg_bits & ((n[c] >= 64 ? ~0ull : ((1ull << n[c]) - 1ull)) << (seg * W)));
    if (__ballot(!in_order || !prefix) == 0ull) {
      const int first = seg * W, last = seg * W + (n[c] > 0 ? n[c] - 1 : 0);
      const double top = __shfl(s, first), bot = __shfl(s, last);
      hi[c] = n[c] > 0 ? top : -(double)INFINITY;
      lo[c] = n[c] > 0 ? bot : (double)INFINITY;
    } else {
      lo[c] = seg_allmin_f64<W>(has ? s : (double)INFINITY);
      hi[c] = seg_allmax_f64<W>(has ? s : -(double)INFINITY);
    }
  }

  // ---- union of ids in first-appearance order ----------------------------------------------
  int U = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (ch[c].k == 0) continue;
    const int U0 = U;
    const bool v = sl < n[c];  // valid entries form a prefix (-1 padding at the tail)
    const long long my = my_uid[c];
    // the union holds an id once: at most one entry matches, so no early exit is needed and the reads of a
    // group of four are independent (the data-dependent loop paid one LDS round trip per entry)
    int f = -1;
    int u_end = 0;  // wave-uniform loop bound: the longest union among the wave's queries
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int ug = __builtin_amdgcn_readlane(U0, g * W);
      u_end = ug > u_end ? ug : u_end;
    }
    for (int u0 = 0; u0 < u_end; u0 += 4) {
      long long e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = uid[(u0 + i) & (W - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i) f = (u0 + i < U0 && e[i] == my) ? u0 + i : f;
    }
    f = v ? f : -1;
    const bool isnew = v && f < 0;
    const unsigned long long m = __ballot(isnew) & seg_bits;
    const unsigned long long lt = m & ((1ull << lane) - 1ull);
    const int idx = isnew ? U + __popcll(lt) : f;
    if (isnew) {
      uid[idx] = my;
      s_pos[seg][0][idx] = -1;
      s_pos[seg][1][idx] = -1;
      s_pos[seg][2][idx] = -1;
    }
    lds_sync();
    if (v) s_pos[seg][c][idx] = sl;
    U += __popcll(m);
    lds_sync();
  }

  // ---- RRF totals, their min / max -------------------------------------------------------------
  const bool wrrf = (P.method == AMDR_FUSE_WRRF);
  const int u = sl;
  const bool act = u < U;
  int pp[3] = {-1, -1, -1};
  double t = 0.0;
  if (act) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pp[c] = s_pos[seg][c][u];
      if (pp[c] >= 0) {
        const double wc = wrrf ? w[c] : 1.0;
        const double v = wc * (1.0 / (double)(P.rrf_k + pp[c] + 1));
        t = t + v;
      }
    }
  }
  FuseCtx X;
  X.rmn = seg_allmin_f64<W>(act ? t : (double)INFINITY);
  X.rmx = seg_allmax_f64<W>(act ? t : -(double)INFINITY);
  X.rdeg = (X.rmx - X.rmn < 1e-12);
  X.wrrf = wrrf;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    X.w[c] = w[c];
    X.lo[c] = lo[c];
    X.hi[c] = hi[c];
  }

  // ---- score, stable descending rank, filter, scatter --------------------------------------------
  double val[AMDR_FUSE_NVALS];
  int mk = 0;
  if (act) {
    fuse_eval(P, X, t, pp, [&](int c, int p) { return s_chs[seg][c][p]; }, val, mk);
    sc[u] = val[AMDR_FV_SCORE];
  }
  lds_sync();
  int kept = 0;
  const size_t obase = (size_t)qi * max_out;
  if (act) {
    const double s = val[AMDR_FV_SCORE];
    int r = 0;
    for (int v2 = 0; v2 < U; ++v2) {
      const double o = sc[v2];
      r += (o > s) || (o == s && v2 < u);
    }
    if (s >= P.min_final_score) kept = 1;
    out_ids[obase + r] = uid[u];
    out_mask[obase + r] = mk;
#pragma unroll
    for (int x = 0; x < AMDR_FUSE_NVALS; ++x) out_vals[(obase + r) * AMDR_FUSE_NVALS + x] = val[x];
  } else if (live && u < max_out) {  // rows past the union: padding
    out_ids[obase + u] = -1;
    out_mask[obase + u] = 0;
#pragma unroll
    for (int x = 0; x < AMDR_FUSE_NVALS; ++x) out_vals[(obase + u) * AMDR_FUSE_NVALS + x] = 0.0;
  }
  kept = __popcll(__ballot(kept != 0) & seg_bits);
  if (live && sl == 0) out_count[qi] = kept;
}

template <int W>
__global__ __launch_bounds__(64) void fuse_packed_kernel(amdr_fuse_params_t P, ChanIn c0, ChanIn c1, ChanIn c2, int nq,
                                                         int max_out, long long* __restrict__ out_ids,
                                                         double* __restrict__ out_vals, int* __restrict__ out_mask,
                                                         int* __restrict__ out_count) {
  FusePre none;
  none.have[0] = none.have[1] = none.have[2] = false;
  fuse_packed_body<W, false>(P, c0, c1, c2, nq, max_out, out_ids, out_vals, out_mask, out_count, none, blockIdx.x * (64 / W));
}

// Dense top-k + fusion in ONE kernel for the serving shape under a batch (dense + BM25, <= 1 024 rows, kd + kb <= 32):
// two queries per wave, lanes 0-31 / 32-63 — the mapping of scores_pair_topk_kernel AND of fuse_packed_kernel<32>.
// The half-wave ranks its row of the score matrix S (the same selector, the same bits), writes the dense channel's
// own (scores, ids) and keeps them in its lanes — lane j = list position j, exactly what the packed fusion wants —
// while the BM25 list it requested BEFORE the selection arrives.  Against the two launches: no store + reload of
// the dense list, one memory round trip of the fusion hidden behind the selection, one launch and one wave start
// fewer per two queries.  Mass ties at the cut (the selector's -1) rank the two rows one after the other with the
// staged selector, as scores_pair_topk_kernel does, and then fuse from its list.
__global__ __launch_bounds__(64) void dense_select_fuse_kernel(amdr_fuse_params_t P, const float* __restrict__ S,
                                                               long ldS, long n, int nq, int kd, int cap,
                                                               float* __restrict__ fin_scores,
                                                               long long* __restrict__ fin_ids, ChanIn c0, ChanIn c1,
                                                               int max_out, long long* __restrict__ out_ids,
                                                               double* __restrict__ out_vals, int* __restrict__ out_mask,
                                                               int* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* buf = reinterpret_cast<C32*>(smem);  // cap entries (>= 128): staged-selector list; the pair selector uses 64
  const int lane = threadIdx.x, sl = lane & 31;
  const int q = 2 * blockIdx.x + (lane >> 5);
  const bool has_q = q < nq;
  FusePre pre;
  pre.have[0] = pre.have[1] = true;
  pre.have[2] = false;
  pre.id[1] = -1;
  pre.s[1] = 0.0;
  if (has_q && sl < c1.k) {  // the BM25 list: in flight during the selection
    pre.id[1] = c1.ids[(size_t)q * c1.k + sl];
    pre.s[1] = chan_score(c1, q, sl);
  }
  C32 out = C32::pad();
  int got = select_row_pair_any(S, ldS, n, q, has_q, kd, lane, buf, out);
  if (got < 0) {  // wave-uniform: mass ties at the cut in one of the two rows
    for (int hh = 0; hh < 2; ++hh) {
      const int qq = 2 * blockIdx.x + hh;
      if (qq >= nq) break;
      const float* row = S + (size_t)qq * ldS;
      WaveTopK<C32> tk;
      tk.init(buf, cap, kd);
      for (long base = 0; base < n; base += 64) {
        const long r = base + lane;
        const bool v = r < n;
        tk.push_lanes(v ? C32::make(row[r], (u32)r) : C32::pad(), v, lane);
      }
      tk.finalize(lane);
      if ((lane >> 5) == hh) {
        got = tk.cnt;
        out = sl < tk.cnt ? tk.buf[sl] : C32::pad();
      }
      wave_lds_fence();
    }
  }
  const bool v = sl < got;
  if (has_q && sl < kd) {
    fin_scores[(size_t)q * kd + sl] = v ? out.score() : -FLT_MAX;
    fin_ids[(size_t)q * kd + sl] = v ? out.id() : -1ll;
  }
  pre.id[0] = v ? out.id() : -1ll;
  pre.s[0] = v ? (double)out.score() : 0.0;
  ChanIn none;
  none.ids = nullptr;
  none.scores = nullptr;
  none.row2uid = nullptr;
  none.k = 0;
  none.is_f64 = 0;
  fuse_packed_body<32, true>(P, c0, c1, none, nq, max_out, out_ids, out_vals, out_mask, out_count, pre, blockIdx.x * 2);
}

// ---- second pass of the two-pass long-batch dense form (round 4; first pass: dense_small_hi.hip) -----------------------
// S holds APPROXIMATE scores (fp16 roundings of both operands, exact products, fp32 sums) and eps[q] the proven bound on
// their distance from the exact dot products.  Per query (two per wave, a half-wave each, as dense_select_fuse_kernel):
//   1. the pair selector's first stage (rows at or above the k-th best lane maximum, sorted: the first k are the k best
//      approximate scores), then the rows at or above (k-th best approximate score) - 2 eps — every row that can be in the
//      exact top-k: a prefix of that list, or one more sweep of the row when the margin reaches below the first threshold;
//   2. their EXACT fp32 dot products, one candidate at a time, the half-wave's 32 lanes across the row (512-byte loads, a
//      butterfly sum);
//   3. sorted by (exact score, lower id first): the dense channel's top-k — then the fusion, as before.
// More than 32 rows inside the margin (mass near-ties), or no bound for the query (eps NaN): the half-wave re-scores EVERY
// row exactly (into LDS) and the plain selectors run on that.  margin_scale (test hook) widens the margin.
template <int V>
__device__ __forceinline__ int select_row_pair_margin(const float* __restrict__ S, long ldS, long n, int q, bool has_q, int k,
                                                      float margin, int lane, C32* scratch, C32& out, int& need) {
  const float* row = S + (size_t)q * ldS;
  const int j = lane & 31;
  tk_v4f blk[V / 4];
#pragma unroll
  for (int u = 0; u < V / 4; ++u) {
    const long c0 = 128L * u + 4 * j;
    const tk_v4f z = {0.f, 0.f, 0.f, 0.f};
    blk[u] = (has_q && c0 < ldS) ? __builtin_nontemporal_load(reinterpret_cast<const tk_v4f*>(row + c0)) : z;  // read once
  }
  u32 sk[V];
#pragma unroll
  for (int u = 0; u < V / 4; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long r = 128L * u + 4 * j + e;
      sk[4 * u + e] = (has_q && r < n) ? ord32(blk[u][e]) : 0u;
    }
  K32 lb;
  lb.c = 0u;
#pragma unroll
  for (int v = 0; v < V; ++v) lb.c = sk[v] > lb.c ? sk[v] : lb.c;
  const K32 sorted_best = wave_sortN_desc<K32, 32>(lb, lane);
  const int kk = (k - 1 < 31 ? k - 1 : 31);
  // (cross-lane reads at wave-uniform positions are two v_readlane and a select; the prefix sum below is five DPP row
  // operations — __shfl / __shfl_up are LDS round trips, eight of them per call of this selector before)
  auto half_lane = [&](int x, int pos) -> int {  // lane `pos` of this lane's half
    const int a = __builtin_amdgcn_readlane(x, pos), b = __builtin_amdgcn_readlane(x, 32 + pos);
    return (lane & 32) ? b : a;
  };
  const u32 T = (u32)half_lane((int)sorted_best.c, kk);  // k-th lane best of this half: <= the k-th best score
  // the rows at or above a key threshold -> this half's 32 scratch slots, sorted into the lanes; -1: more than 32
  auto gather = [&](u32 Te, C32& c) -> int {
    int mine = 0;
#pragma unroll
    for (int v = 0; v < V; ++v) mine += (sk[v] >= Te) ? 1 : 0;
    int incl = mine;  // inclusive prefix sum over the 32 lanes of the half (rows of 16, then row 0 -> 1, 2 -> 3)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);  // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);  // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);  // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);  // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
    const int cnt = half_lane(incl, 31);
    if (cnt > 32) return -1;
    int at = (lane & 32) + incl - mine;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (sk[v] >= Te) {
        C32 e;
        e.c = ((u64)sk[v] << 32) | (u64)(0xffffffffu - (u32)(128 * (v >> 2) + 4 * j + (v & 3)));
        scratch[at++] = e;
      }
    }
    wave_lds_fence();
    c = (j < cnt) ? scratch[lane] : C32::pad();
    c = wave_sortN_desc<C32, 32>(c, lane);
    wave_lds_fence();
    return cnt;
  };
  // stage 1, the exact form's own selection: the rows at or above the k-th LANE maximum (a few more than k: its first k
  // are the k best approximate scores).  With the margin already taken off that threshold (the first version), depths from
  // ~14 up had more than 32 survivors on most queries and fell back to re-scoring the whole row: 732 against 559 us at
  // k = 16, 3.4 against 0.6 ms at k = 20 (1 024 x 768, 37 376 queries).
  const u32 Te1 = T > 1u ? T : 1u;  // (T == 0: fewer than k rows — every row)
  C32 c;
  int cnt = gather(Te1, c);
  if (cnt < 0) return -1;  // (this lane's half; the caller votes)
  // stage 2: everything at or above (k-th best approximate score) - margin.  Usually a prefix of the sorted survivors;
  // when the margin reaches below the lane-maximum threshold the row is swept again with the cut itself.
  u32 cut = 1u;
  if (cnt > kk) {
    const float tk_f = __int_as_float(half_lane(__float_as_int(c.score()), kk));
    cut = ord32(tk_f - margin);
    cut = cut > 1u ? cut : 1u;
  }
  if (cut < Te1) {  // (half-uniform)
    cnt = gather(cut, c);
    if (cnt < 0) return -1;
    need = cnt;
  } else {
    const unsigned long long m = __ballot(j < cnt && (u32)(c.c >> 32) >= cut);
    need = __popcll((lane & 32) ? (m >> 32) : (m & 0xffffffffull));
  }
  out = c;
  return cnt;
}

template <bool FUSE, int D128>  // D128 = d / 128 (the row a half-wave re-scores: D128 16-byte pieces per lane)
__global__ __launch_bounds__(64) void dense_hi_select_fuse_kernel(amdr_fuse_params_t P, const float* __restrict__ S, long ldS, long n,
                                                                  int nq, int kd, const float* __restrict__ X,
                                                                  const float* __restrict__ Q, int d,
                                                                  const float* __restrict__ eps, float margin_scale,
                                                                  float* __restrict__ fin_scores, long long* __restrict__ fin_ids,
                                                                  ChanIn c0, ChanIn c1, int max_out,
                                                                  long long* __restrict__ out_ids, double* __restrict__ out_vals,
                                                                  int* __restrict__ out_mask, int* __restrict__ out_count,
                                                                  unsigned int* __restrict__ fallbacks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* buf = reinterpret_cast<C32*>(smem);  // 128 entries: the selectors' scratch
  const int lane = threadIdx.x, sl = lane & 31, half = lane >> 5;
  const int q = 2 * blockIdx.x + half;
  const bool has_q = q < nq;
  FusePre pre;
  pre.have[0] = pre.have[1] = true;
  pre.have[2] = false;
  pre.id[1] = -1;
  pre.s[1] = 0.0;
  if (FUSE && has_q && sl < c1.k) {  // the BM25 list: in flight during the selection
    pre.id[1] = c1.ids[(size_t)q * c1.k + sl];
    pre.s[1] = chan_score(c1, q, sl);
  }
  const float e_q = has_q ? eps[q] : 0.f;
  const float margin = 2.f * e_q * margin_scale;
  bool exact_all = has_q && !(margin == margin && margin <= FLT_MAX);  // no bound for this query
  // this half's query, spread over its 32 lanes: lane sl holds components 128 u + 4 sl .. + 3
  tk_v4f qv[D128];
#pragma unroll
  for (int u = 0; u < D128; ++u) {
    const tk_v4f z = {0.f, 0.f, 0.f, 0.f};
    qv[u] = has_q ? __builtin_nontemporal_load(reinterpret_cast<const tk_v4f*>(Q + (size_t)q * d + 128 * u + 4 * sl)) : z;  // (the
    // streamed score rows and queries are read once: non-temporal, so that the chunk rows the candidates re-read stay in L2)
  }
  // the sum of a value over the 32 lanes of each half, in every lane of the half: five DPP row operations leave the halves'
  // sums in lanes 31 and 63 (dense_dot.hpp), two v_readlane hand them out — no LDS round trips (ds_bpermute shuffles made
  // the re-scoring a chain of ~10 of them per candidate)
  auto half_sum = [&](float v) -> float {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return half ? s1 : s0;
  };
  auto row_load = [&](long r, tk_v4f (&xv)[D128]) {
    const float* xr = X + (size_t)r * d + 4 * sl;
#pragma unroll
    for (int u = 0; u < D128; ++u) xv[u] = *reinterpret_cast<const tk_v4f*>(xr + 128 * u);
  };
  auto row_fma = [&](const tk_v4f (&xv)[D128]) -> float {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < D128; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(xv[u][e], qv[u][e], acc);
    return half_sum(acc);
  };
  auto row_dot = [&](long r) -> float {  // every lane of the half returns <Q[q], X[r]> (r: uniform in the half)
    tk_v4f xv[D128];
    row_load(r, xv);
    return row_fma(xv);
  };
  C32 out = C32::pad();
  int need = 0, got = 0;
  {
    // (a half without a bound runs the selector on whatever its row holds and discards the result: the other half of the
    // wave needs its own)
    const float mg = (has_q && !exact_all) ? margin : 0.f;
    if (n <= 256)
      got = select_row_pair_margin<8>(S, ldS, n, q, has_q, kd, mg, lane, buf, out, need);
    else if (n <= 512)
      got = select_row_pair_margin<16>(S, ldS, n, q, has_q, kd, mg, lane, buf, out, need);
    else if (n <= 640)
      got = select_row_pair_margin<20>(S, ldS, n, q, has_q, kd, mg, lane, buf, out, need);
    else
      got = select_row_pair_margin<32>(S, ldS, n, q, has_q, kd, mg, lane, buf, out, need);
    exact_all = exact_all || (has_q && got < 0);
    if (exact_all) need = 0;
  }
  if (__any(exact_all)) {  // wave-uniform: one of the two halves (or both) re-scores its whole row
    const unsigned long long fbm = __ballot(exact_all && sl == 0);  // (one lane per half)
    if (lane == 0 && fallbacks) atomicAdd(fallbacks, (unsigned int)__popcll(fbm));
    const C32 keep = out;
    const int keep_need = need, keep_got = got;
    // the exact scores of the half's whole row, in LDS (two rows of 1 024 floats behind the selectors' scratch), then the
    // plain selectors on them
    float* xs = reinterpret_cast<float*>(buf + 128);
    for (int r = sl; r < 1024; r += 32) xs[1024 * half + r] = 0.f;
    wave_lds_fence();
    for (long r = 0; r < n; ++r) {
      const float v = row_dot(r);
      if (exact_all && sl == 0) xs[1024 * half + r] = v;
    }
    wave_lds_fence();
    C32 o2 = C32::pad();
    int g2 = select_row_pair_any(xs, 1024, n, half, has_q && exact_all, kd, lane, buf, o2);
    if (g2 < 0) {  // mass ties at the cut among EXACT scores: the staged selector, one half after the other
      for (int hh = 0; hh < 2; ++hh) {
        const int qq = 2 * blockIdx.x + hh;
        const bool mine_h = __shfl((int)exact_all, 32 * hh) != 0;
        if (qq >= nq || !mine_h) continue;
        const float* row = xs + 1024 * hh;
        WaveTopK<C32> tk;
        tk.init(buf, 128, kd);
        for (long base = 0; base < n; base += 64) {
          const long r = base + lane;
          const bool v = r < n;
          tk.push_lanes(v ? C32::make(row[r], (u32)r) : C32::pad(), v, lane);
        }
        tk.finalize(lane);
        if (half == hh) {
          g2 = tk.cnt;
          o2 = sl < tk.cnt ? tk.buf[sl] : C32::pad();
        }
        wave_lds_fence();
      }
    }
    if (exact_all) {  // this half's list is final: exact scores already
      out = o2;
      got = g2 < kd ? g2 : kd;
      need = 0;
    } else {
      out = keep;
      got = keep_got;
      need = keep_need;
    }
  }
  // ---- exact scores of the candidates (the first `need` survivors of each half), one per step
  const int steps = __builtin_amdgcn_readfirstlane(max(__shfl(need, 0), __shfl(need, 32)));
  float mine_exact = 0.f;
  const int my_id = (int)out.id();
  // two candidates per step: both rows requested before either is summed.  (Tried: the NEXT step's rows requested before
  // this step's are summed, two register sets in ping-pong — 146 VGPRs, three waves per SIMD instead of four: 119.5 against
  // 119.1 us; the kernel issues 2 289 vector instructions per wave = 56 % of its time and waits on memory for half of it.
  // The other direction, amdgpu_waves_per_eu(5) / (6): registers capped at 96 / 80, the rest spilled (4 / 172 at d = 384, more at 768) — the d = 768 step 0.258 -> 0.279 / 0.291 ms.)
  for (int c = 0; c < steps; c += 2) {  // (steps: wave-uniform)
    // candidates c, c + 1 of each half: their lanes are wave-uniform (v_readlane), the half picks its own
    const int c1 = c + 1 < 32 ? c + 1 : 31;
    const int ra0 = __builtin_amdgcn_readlane(my_id, c), ra1 = __builtin_amdgcn_readlane(my_id, 32 + c);
    const int rb0 = __builtin_amdgcn_readlane(my_id, c1), rb1 = __builtin_amdgcn_readlane(my_id, 32 + c1);
    const bool la = c < need, lb = c + 1 < need;
    tk_v4f xa[D128], xb[D128];
    row_load(la ? (long)(half ? ra1 : ra0) : 0, xa);
    row_load(lb ? (long)(half ? rb1 : rb0) : 0, xb);
    const float va = row_fma(xa), vb = row_fma(xb);
    if (la && sl == c) mine_exact = va;
    if (lb && sl == c + 1) mine_exact = vb;
  }
  if (need > 0) {
    C32 c = (sl < need) ? C32::make(mine_exact, (u32)out.id()) : C32::pad();
    c = wave_sortN_desc<C32, 32>(c, lane);
    out = c;
    got = need < kd ? need : kd;
  }
  const bool v = sl < got && sl < kd;
  if (has_q && sl < kd) {
    fin_scores[(size_t)q * kd + sl] = v ? out.score() : -FLT_MAX;
    fin_ids[(size_t)q * kd + sl] = v ? out.id() : -1ll;
  }
  if (FUSE) {
    pre.id[0] = v ? out.id() : -1ll;
    pre.s[0] = v ? (double)out.score() : 0.0;
    ChanIn none;
    none.ids = nullptr;
    none.scores = nullptr;
    none.row2uid = nullptr;
    none.k = 0;
    none.is_f64 = 0;
    fuse_packed_body<32, true>(P, c0, c1, none, nq, max_out, out_ids, out_vals, out_mask, out_count, pre, blockIdx.x * 2);
  }
}

// ---- the serving call in ONE launch ------------------------------------------------------------------------------------
// HybridRetriever.search() issues one query at a time (hybrid_retriever.py:282-384); on a serving corpus (591 / 1 260
// chunks) its dense + BM25 step was FOUR short launches — BM25 scoring + top-k, one wave per (query, row) of dense scores,
// the dense top-k, the fusion — 45 us of which ~15 are kernels.  Here one launch does all four for 1-4 queries on a
// corpus of <= 2 048 chunks: blocks take ROLES — block 0 of a query is its BM25 wave (bm25_core.hpp bm25_block_query: the
// channel's own code), blocks 1.. take 16 chunk rows each (dense_dot.hpp dense_row_dot: one wave per row, the GEMV
// form's instruction sequence) — and hand over through two self-resetting arrival counters per query (a wave
// drains its stores, releases at agent scope and takes a ticket; the last ticket holder acquires — MI355X_MICROARCH.md,
// inter-workgroup visibility): the LAST dense block to arrive ranks the score row (the register selector of the slab
// top-k) while the BM25 wave is still scoring, and the SECOND of the two finished channel lists to arrive fuses
// (fuse_packed_body, the packed fusion's code).  The same instructions as the four launches, hence the same bits
// (tests/test_hybrid_small_gpu.py).
struct SmallArgs {
  // BM25 role
  const long long* term_ptr;
  const int* post_doc;
  const double* post_w;
  const double* idf;
  long n_terms, n_docs;
  const int* q_terms;
  const long long* q_ptr;
  int kb, cap, slab, use_select;
  double* bm_scores;     // [nq, kb]
  long long* bm_ids;
  // dense role
  const float* X;
  const float* Q;
  long n_rows;
  int d, rows_per_block, blocks_per_query;
  float* S;              // [nq, ldS]
  long ldS;
  int kd, cap_sel;
  float* d_scores;       // [nq, kd]
  long long* d_ids;
  int* ticket;           // [>= nq] zero before the launch, zero after
};

// one wave's arrival at a counter: its stores are out and released at agent scope; returns the ticket (wave-uniform)
__device__ __forceinline__ int small_arrive(int* counter, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int t = 0;
  if (lane == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __builtin_amdgcn_readfirstlane(t);
}
__device__ __forceinline__ void small_acquire(int* counter, int lane) {
  if (lane == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NVT>
__global__ __launch_bounds__(256) void hybrid_small_kernel(SmallArgs A, amdr_fuse_params_t P, ChanIn c0, ChanIn c1, int nq,
                                                           int max_out, long long* __restrict__ out_ids,
                                                           double* __restrict__ out_vals, int* __restrict__ out_mask,
                                                           int* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int q = blockIdx.y, role = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* rows_done = A.ticket + q;       // arrivals of the dense blocks of query q
  int* lists_done = A.ticket + 32 + q; // arrivals of its two channel lists
  const int sl = lane & 31;
  FusePre pre;
  pre.have[0] = pre.have[1] = true;
  pre.have[2] = false;
  bool have_dense = false;  // this wave ranked the dense row itself: the list is in its lanes
  if (role == 0) {
    // ---- the BM25 channel of query q: one wave, the channel's own code; the other three have nothing to do
    if (wave != 0) return;
    bm25_block_query<1, NVT>(A.term_ptr, A.post_doc, A.post_w, A.idf, A.n_terms, A.n_docs, A.q_terms, A.q_ptr, nq, A.kb,
                             A.cap, A.slab, A.use_select, nullptr, nullptr, A.bm_scores, A.bm_ids, q, 0, smem);
  } else {
    // ---- rows_per_block chunk rows of the dense channel, one wave per row (the GEMV form's dot product)
    const long r0 = (long)(role - 1) * A.rows_per_block;
    long r1 = r0 + A.rows_per_block;
    if (r1 > A.n_rows) r1 = A.n_rows;
    for (long r = r0 + wave; r < r1; r += 4) {
      const float acc = dense_row_dot(A.X + (size_t)r * A.d, A.Q + (size_t)q * A.d, A.d, lane);
      if (lane == 63) A.S[(size_t)q * A.ldS + r] = acc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's scores are out
    if (wave != 0) return;
    if (small_arrive(rows_done, lane) != A.blocks_per_query - 2) return;
    // ---- the last dense block to arrive: top-k of row q (scores_slab_topk_kernel<1>'s selection)
    small_acquire(rows_done, lane);
    C32* buf = reinterpret_cast<C32*>(smem);
    const float* row = A.S + (size_t)q * A.ldS;
    const long n = A.n_rows;
    WaveTopK<C32> tk;
    tk.init(buf, A.cap_sel, A.kd);
    int got;
    if (n <= 256)
      got = select_row<4>(row, 0, n, A.kd, lane, tk.buf);
    else if (n <= 640)
      got = select_row<10>(row, 0, n, A.kd, lane, tk.buf);
    else if (n <= 1024)
      got = select_row<16>(row, 0, n, A.kd, lane, tk.buf);
    else if (n <= 1280)
      got = select_row<20>(row, 0, n, A.kd, lane, tk.buf);
    else
      got = select_row<32>(row, 0, n, A.kd, lane, tk.buf);
    if (got >= 0) {
      tk.cnt = got;
    } else {  // mass ties at the cut: the staged selector
      for (long base = 0; base < n; base += 64) {
        const long r = base + lane;
        const bool v = r < n;
        tk.push_lanes(v ? C32::make(row[r], (u32)r) : C32::pad(), v, lane);
      }
      tk.finalize(lane);
    }
    wave_lds_fence();
    const bool v = lane < 32 && sl < tk.cnt;
    const C32 mine = v ? tk.buf[sl] : C32::pad();
    if (lane < A.kd) {
      A.d_scores[(size_t)q * A.kd + lane] = v ? mine.score() : -FLT_MAX;
      A.d_ids[(size_t)q * A.kd + lane] = v ? mine.id() : -1ll;
    }
    pre.id[0] = v ? mine.id() : -1ll;
    pre.s[0] = v ? (double)mine.score() : 0.0;
    have_dense = true;
    wave_lds_fence();
  }
  // ---- a finished channel list; the second of the two to arrive fuses (the BM25 wave while the dense rows were being
  // ranked elsewhere, or the ranking wave when BM25 finished first)
  if (small_arrive(lists_done, lane) != 1) return;
  small_acquire(lists_done, lane);
  if (!have_dense) {
    pre.id[0] = -1;
    pre.s[0] = 0.0;
    if (lane < 32 && sl < A.kd) {
      const long long id = A.d_ids[(size_t)q * A.kd + sl];
      pre.id[0] = id;
      pre.s[0] = id >= 0 ? (double)A.d_scores[(size_t)q * A.kd + sl] : 0.0;
    }
  }
  pre.id[1] = -1;
  pre.s[1] = 0.0;
  if (lane < 32 && sl < c1.k) {
    pre.id[1] = c1.ids[(size_t)q * c1.k + sl];
    pre.s[1] = chan_score(c1, q, sl);
  }
  ChanIn none;
  none.ids = nullptr;
  none.scores = nullptr;
  none.row2uid = nullptr;
  none.k = 0;
  none.is_f64 = 0;
  // segment 0 (lanes 0-31) = query q, segment 1 has no query (q + 1 >= the limit handed in)
  fuse_packed_body<32, true>(P, c0, c1, none, q + 1, max_out, out_ids, out_vals, out_mask, out_count, pre, q);
}

// fuse_kernel for long candidate lists, the packed forms when a query fits in 32 or 16 lanes
static void launch_fuse(const amdr_fuse_params_t& P, const ChanIn& c0, const ChanIn& c1, const ChanIn& c2, int nq,
                        int max_out, long long* ids, double* vals, int* mask, int* count, hipStream_t st) {
  if (max_out <= 16)
    hipLaunchKernelGGL(fuse_packed_kernel<16>, dim3((nq + 3) / 4), dim3(64), 0, st, P, c0, c1, c2, nq, max_out, ids, vals,
                       mask, count);
  else if (max_out <= 32)
    hipLaunchKernelGGL(fuse_packed_kernel<32>, dim3((nq + 1) / 2), dim3(64), 0, st, P, c0, c1, c2, nq, max_out, ids, vals,
                       mask, count);
  else
    hipLaunchKernelGGL(fuse_kernel, dim3(nq), dim3(64), (size_t)max_out * 36, st, P, c0, c1, c2, max_out, ids, vals, mask,
                       count);
}

// block = 64 threads; grid = nq.  Dynamic LDS: staging for one query's lists.
__global__ __launch_bounds__(64) void rerank_blend_kernel(int max_out, const int* __restrict__ count,
                                                          long long* __restrict__ ids, double* __restrict__ vals,
                                                          int* __restrict__ mask, const double* __restrict__ ce_raw,
                                                          int top_n, double beta, double* __restrict__ out_rerank) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* svals = reinterpret_cast<double*>(smem);                   // [max_out][NVALS]
  double* snew = svals + (size_t)max_out * AMDR_FUSE_NVALS;            // new score per input position
  double* snorm = snew + max_out;                                    // norm per candidate
  long long* sids = reinterpret_cast<long long*>(snorm + max_out);   // [max_out]
  int* smask = reinterpret_cast<int*>(sids + max_out);               // [max_out]
  int* spre = smask + max_out;                                       // position in the pre-sort sequence
  const int lane = threadIdx.x, qi = blockIdx.x;
  const size_t base = (size_t)qi * max_out;
  const int cnt = count[qi];
  const int n = cnt < top_n ? cnt : top_n;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);

  for (int r = lane; r < max_out; r += 64) {
    out_rerank[(base + r) * 2 + 0] = nan;
    out_rerank[(base + r) * 2 + 1] = nan;
  }
  if (n <= 0) return;

  double mn = INFINITY, mx = -INFINITY;
  for (int j = lane; j < n; j += 64) {
    double r = ce_raw[(size_t)qi * top_n + j];
    mn = fmin(mn, r);
    mx = fmax(mx, r);
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  const bool deg = (mx - mn < 1e-12);

  for (int j = lane; j < cnt; j += 64) {
    for (int x = 0; x < AMDR_FUSE_NVALS; ++x) svals[(size_t)j * AMDR_FUSE_NVALS + x] = vals[(base + j) * AMDR_FUSE_NVALS + x];
    sids[j] = ids[base + j];
    smask[j] = mask[base + j];
    double s = svals[(size_t)j * AMDR_FUSE_NVALS + AMDR_FV_SCORE];
    if (j < n) {
      double raw = ce_raw[(size_t)qi * top_n + j];
      double nm = deg ? 0.0 : (raw - mn) / (mx - mn);
      snorm[j] = nm;
      s = (1 - beta) * s + beta * nm;
    }
    snew[j] = s;
  }
  lds_sync();
  // pre-sort sequence: candidates by norm desc (stable), then the untouched tail
  for (int j = lane; j < cnt; j += 64) {
    int p = j;
    if (j < n) {
      const double nm = snorm[j];
      p = 0;
      for (int i = 0; i < n; ++i) {
        const double o = snorm[i];
        p += (o > nm) || (o == nm && i < j);
      }
    }
    spre[j] = p;
  }
  lds_sync();
  for (int j = lane; j < cnt; j += 64) {
    const double s = snew[j];
    const int pj = spre[j];
    int r = 0;
    for (int i = 0; i < cnt; ++i) {
      const double o = snew[i];
      r += (o > s) || (o == s && spre[i] < pj);
    }
    ids[base + r] = sids[j];
    mask[base + r] = smask[j];
    for (int x = 0; x < AMDR_FUSE_NVALS; ++x) vals[(base + r) * AMDR_FUSE_NVALS + x] = svals[(size_t)j * AMDR_FUSE_NVALS + x];
    vals[(base + r) * AMDR_FUSE_NVALS + AMDR_FV_SCORE] = s;
    if (j < n) {
      out_rerank[(base + r) * 2 + 0] = ce_raw[(size_t)qi * top_n + j];
      out_rerank[(base + r) * 2 + 1] = snorm[j];
    }
  }
}

static size_t rerank_lds(int max_out) {
  return (size_t)max_out * (AMDR_FUSE_NVALS + 2) * sizeof(double) + (size_t)max_out * sizeof(long long) +
         (size_t)max_out * 2 * sizeof(int);
}

bool dense_select_fuse_applies(long n, int slabs, int m, int kd, int kb) {
  const char* e = getenv("AMDR_DENSE_FUSE");  // "0" pins the two-launch form (A/B, tests)
  if (e && e[0] == '0') return false;
  return slabs == 1 && n >= 1 && n <= 1024 && kd >= 1 && kd <= 32 && kb >= 0 && kd + kb <= 32 && m >= 1;
}

int dense_select_fuse_launch(const FuseTail& t, int q0, const float* S, long ldS, long n, int m, int kd, int cap,
                             float* fin_scores, int64_t* fin_ids, hipStream_t st) {
  const int mo = kd + t.kb;
  ChanIn c0{nullptr, nullptr, (const long long*)t.dense_row2uid, kd, 0};
  ChanIn c1{(const long long*)(t.kb ? t.bm25_ids + (size_t)q0 * t.kb : nullptr),
            t.kb ? (const void*)(t.bm25_scores + (size_t)q0 * t.kb) : nullptr, (const long long*)t.bm25_row2uid, t.kb, 1};
  hipLaunchKernelGGL(dense_select_fuse_kernel, dim3((m + 1) / 2), dim3(64), (size_t)cap * sizeof(C32), st, *t.p, S, ldS, n,
                     m, kd, cap, fin_scores, (long long*)fin_ids, c0, c1, mo, (long long*)(t.out_ids + (size_t)q0 * mo),
                     t.out_vals + (size_t)q0 * mo * AMDR_FUSE_NVALS, t.out_mask + (size_t)q0 * mo, t.out_count + q0);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

// second pass of the two-pass long-batch form (dense_hi_select_fuse_kernel); t == nullptr: the dense lists only
int dense_hi_select_launch(const FuseTail* t, int q0, const float* S, long ldS, long n, int m, int kd, const float* X,
                           const float* Q, int d, const float* eps, float* fin_scores, int64_t* fin_ids,
                           unsigned int* fallbacks, hipStream_t st) {
  const char* ms = getenv("AMDR_DENSE_SMALL_HI_MARGIN");  // test hook: widens the candidate margin (a huge one: every
  const float margin_scale = ms ? (float)atof(ms) : 1.f;  // query takes the exact fallback inside the kernel)
  const size_t lds = 128 * sizeof(C32) + 2 * 1024 * sizeof(float);  // the selectors' scratch + two rows of exact scores
  if (t) {
    const int mo = kd + t->kb;
    ChanIn c0{nullptr, nullptr, (const long long*)t->dense_row2uid, kd, 0};
    ChanIn c1{(const long long*)(t->kb ? t->bm25_ids + (size_t)q0 * t->kb : nullptr),
              t->kb ? (const void*)(t->bm25_scores + (size_t)q0 * t->kb) : nullptr, (const long long*)t->bm25_row2uid, t->kb, 1};
#define AMDR_HSF_FUSED(D)                                                                                                  \
  hipLaunchKernelGGL((dense_hi_select_fuse_kernel<true, D>), dim3((m + 1) / 2), dim3(64), lds, st, *t->p, S, ldS, n, m, kd, X, \
                     Q, d, eps, margin_scale, fin_scores, (long long*)fin_ids, c0, c1, mo,                                 \
                     (long long*)(t->out_ids + (size_t)q0 * mo), t->out_vals + (size_t)q0 * mo * AMDR_FUSE_NVALS,          \
                     t->out_mask + (size_t)q0 * mo, t->out_count + q0, fallbacks)
    switch (d >> 7) {
      case 1: AMDR_HSF_FUSED(1); break;
      case 2: AMDR_HSF_FUSED(2); break;
      case 3: AMDR_HSF_FUSED(3); break;
      case 4: AMDR_HSF_FUSED(4); break;
      case 5: AMDR_HSF_FUSED(5); break;
      case 6: AMDR_HSF_FUSED(6); break;
      case 7: AMDR_HSF_FUSED(7); break;
      default: AMDR_HSF_FUSED(8); break;
    }
#undef AMDR_HSF_FUSED
  } else {
    amdr_fuse_params_t P{};
    ChanIn none{nullptr, nullptr, nullptr, 0, 0};
#define AMDR_HSF_PLAIN(D)                                                                                                   \
  hipLaunchKernelGGL((dense_hi_select_fuse_kernel<false, D>), dim3((m + 1) / 2), dim3(64), lds, st, P, S, ldS, n, m, kd, X, Q, \
                     d, eps, margin_scale, fin_scores, (long long*)fin_ids, none, none, 0, (long long*)nullptr,             \
                     (double*)nullptr, (int*)nullptr, (int*)nullptr, fallbacks)
    switch (d >> 7) {
      case 1: AMDR_HSF_PLAIN(1); break;
      case 2: AMDR_HSF_PLAIN(2); break;
      case 3: AMDR_HSF_PLAIN(3); break;
      case 4: AMDR_HSF_PLAIN(4); break;
      case 5: AMDR_HSF_PLAIN(5); break;
      case 6: AMDR_HSF_PLAIN(6); break;
      case 7: AMDR_HSF_PLAIN(7); break;
      default: AMDR_HSF_PLAIN(8); break;
    }
#undef AMDR_HSF_PLAIN
  }
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int dense_fuse_plain_launch(const FuseTail& t, int q0, int m, int kd, const float* dense_scores, const int64_t* dense_ids,
                            hipStream_t st) {
  const int mo = kd + t.kb;
  ChanIn c0{(const long long*)dense_ids, dense_scores, (const long long*)t.dense_row2uid, kd, 0};
  ChanIn c1{(const long long*)(t.kb ? t.bm25_ids + (size_t)q0 * t.kb : nullptr),
            t.kb ? (const void*)(t.bm25_scores + (size_t)q0 * t.kb) : nullptr, (const long long*)t.bm25_row2uid, t.kb, 1};
  ChanIn c2{nullptr, nullptr, nullptr, 0, 0};
  launch_fuse(*t.p, c0, c1, c2, m, mo, (long long*)(t.out_ids + (size_t)q0 * mo),
              t.out_vals + (size_t)q0 * mo * AMDR_FUSE_NVALS, t.out_mask + (size_t)q0 * mo, t.out_count + q0, st);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

// AMDR_HYBRID_SMALL=0 pins the separate launches (A/B and tests)
bool hybrid_small_applies(long n_dense, long n_bm25, int nslabs, int nq, int kd, int kb) {
  const char* e = getenv("AMDR_HYBRID_SMALL");
  if (e && e[0] == '0') return false;
  return nq >= 1 && nq <= 4 && n_dense >= 1 && n_dense <= kSelectRowsMax && n_bm25 >= 1 && nslabs == 1 && kd >= 1 && kb >= 1 &&
         kd + kb <= 32;
}

// chunk rows per dense block (4 waves).  Every block costs an arrival (one atomic on the query's counter) and a block
// start; measured on 591 x 384 ... 2 048 x 768, 1-4 queries (scripts/ab_hybrid_small.py): 16 rows up to ~2 500
// (query, row) pairs, 32 beyond.  AMDR_HYBRID_SMALL_ROWS pins a value (multiple of 4).
static int hybrid_small_rows(long n, int nq) {
  static const int pinned = [] {
    const char* e = getenv("AMDR_HYBRID_SMALL_ROWS");
    int r = e ? atoi(e) : 0;
    if (r <= 0) return 0;
    if (r < 4) r = 4;
    if (r > 64) r = 64;
    return (r + 3) / 4 * 4;
  }();
  if (pinned) return pinned;
  return n * nq <= 2560 ? 16 : 32;
}

int hybrid_small_launch(const DenseRaw& dr, const Bm25Raw& br, const float* Q, const int* q_terms, const long long* q_ptr,
                        int nq, int kd, int kb, const amdr_fuse_params_t& P, const int64_t* dense_row2uid,
                        const int64_t* bm25_row2uid, float* dense_scores, int64_t* dense_ids, double* bm25_scores,
                        int64_t* bm25_ids, int64_t* out_ids, double* out_vals, int32_t* out_mask, int32_t* out_count,
                        hipStream_t st) {
  SmallArgs A;
  A.term_ptr = br.term_ptr;
  A.post_doc = br.post_doc;
  A.post_w = br.post_w;
  A.idf = br.idf;
  A.n_terms = br.n_terms;
  A.n_docs = br.n_docs;
  A.q_terms = q_terms;
  A.q_ptr = q_ptr;
  A.kb = kb;
  A.cap = br.cap;
  A.slab = br.slab;
  A.use_select = br.select_on ? 1 : 0;
  A.bm_scores = bm25_scores;
  A.bm_ids = (long long*)bm25_ids;
  A.X = dr.X;
  A.Q = Q;
  A.d = dr.d;
  A.n_rows = dr.n;
  A.rows_per_block = hybrid_small_rows(dr.n, nq);
  const int dense_blocks = (int)((dr.n + A.rows_per_block - 1) / A.rows_per_block);
  A.blocks_per_query = 1 + dense_blocks;
  A.S = dr.S;
  A.ldS = dr.ld;
  A.kd = kd;
  int cap_sel = topk_cap(kd);
  if (cap_sel < 128) cap_sel = 128;
  A.cap_sel = cap_sel;
  A.d_scores = dense_scores;
  A.d_ids = (long long*)dense_ids;
  A.ticket = br.ticket;
  size_t lds = br.lds;
  if (lds < (size_t)cap_sel * sizeof(C32)) lds = (size_t)cap_sel * sizeof(C32);
  const int mo = kd + kb;
  ChanIn c0{nullptr, nullptr, (const long long*)dense_row2uid, kd, 0};
  ChanIn c1{(const long long*)bm25_ids, (const void*)bm25_scores, (const long long*)bm25_row2uid, kb, 1};
#define AMDR_HS_LAUNCH(NVT)                                                                                        \
  hipLaunchKernelGGL((hybrid_small_kernel<NVT>), dim3(A.blocks_per_query, nq), dim3(256), lds, st, A, P, c0, c1, nq, mo, \
                     (long long*)out_ids, out_vals, out_mask, out_count)
  const int nv = br.nvt;  // (bm_run's choice of register bucket)
  if (nv <= 4) AMDR_HS_LAUNCH(4);
  else if (nv <= 8) AMDR_HS_LAUNCH(8);
  else if (nv <= 10) AMDR_HS_LAUNCH(10);
  else if (nv <= 16) AMDR_HS_LAUNCH(16);
  else if (nv <= 20) AMDR_HS_LAUNCH(20);
  else AMDR_HS_LAUNCH(32);
#undef AMDR_HS_LAUNCH
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

}  // namespace amdr

using namespace amdr;

namespace {

int fuse_check(const amdr_fuse_params_t* p, int nq, int kd, int kb, int kc) {
  AMDR_REQUIRE(p != nullptr, "fuse: null params");
  AMDR_REQUIRE(p->method >= 0 && p->method <= 3, "fuse: unknown method %d", p->method);
  AMDR_REQUIRE(nq >= 0, "fuse: nq=%d", nq);
  AMDR_REQUIRE(kd >= 0 && kd <= AMDR_MAX_K && kb >= 0 && kb <= AMDR_MAX_K && kc >= 0 && kc <= AMDR_MAX_K,
               "fuse: channel depth outside [0,%d]", AMDR_MAX_K);
  AMDR_REQUIRE(kd + kb + kc >= 1, "fuse: all channels empty (max_out would be 0)");
  return AMDR_OK;
}

}  // namespace

// ---- host-pointer conveniences (single-query API path) ---------------------
namespace {
// The host-pointer entry points stage through ONE grow-only device arena per calling thread
// (ten hipMalloc/hipFree pairs per single-query call used to cost more than the kernel).
struct Arena {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  int device = -1;
  ~Arena() {
    if (base) (void)hipFree(base);
  }
  int begin(int dev, size_t total) {
    if (dev != device || total > cap) {
      if (base) (void)hipFree(base);
      base = nullptr;
      cap = 0;
      size_t want = total < (1u << 20) ? (1u << 20) : total;
      AMDR_HIP(hipMalloc((void**)&base, want));
      cap = want;
      device = dev;
    }
    used = 0;
    return AMDR_OK;
  }
  void* take(size_t bytes) {
    void* p = base + used;
    used += (bytes + 255) & ~(size_t)255;
    return p;
  }
};
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
Arena& arena() {
  static thread_local Arena a;
  return a;
}
}  // namespace

// Host-pointer forms: inputs are packed into ONE pinned-size-agnostic host block and moved
// with one H2D copy, outputs come back with one D2H copy (every extra small copy from
// pageable memory costs ~10-15 us).
namespace {
struct Pack {
  std::vector<char>& buf;
  size_t used = 0;
  explicit Pack(std::vector<char>& b) : buf(b) {}
  size_t add(const void* src, size_t bytes) {
    size_t off = used;
    used += pad256(bytes);
    if (buf.size() < used) buf.resize(used);
    if (src && bytes) memcpy(buf.data() + off, src, bytes);
    return off;
  }
};
std::vector<char>& host_block() {
  static thread_local std::vector<char> b;
  return b;
}
}  // namespace

namespace amdr {
// The columns a bulk caller reads, compacted on the device so that ONE small copy serves the host API: the first `w`
// fused hits of every query as rows / scores / channel masks (entries past min(count, w): -1 / 0 / 0) + the clipped
// counts.  The full fused record is 9 doubles per candidate and channel-depth x channels candidates per query — 1.6 KB
// per query at top-10 of two channels; PCIe, not the kernels, then bounds a bulk search (hybrid_retriever
// search_batch_arrays: 15 MB per 9 344 queries).  Here: 20 bytes per kept hit.
__global__ __launch_bounds__(256) void fuse_compact_kernel(const long long* __restrict__ ids, const double* __restrict__ vals,
                                                           const int* __restrict__ mask, const int* __restrict__ count,
                                                           int nq, int max_out, int w, long long* __restrict__ out_rows,
                                                           double* __restrict__ out_scores, int* __restrict__ out_mask,
                                                           int* __restrict__ out_count) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nq * w) return;
  const int q = (int)(i / w), j = (int)(i - (long)q * w);
  const int c = count[q] < w ? count[q] : w;
  const bool keep = j < c;
  const size_t src = (size_t)q * max_out + j;
  out_rows[i] = keep ? ids[src] : -1ll;
  out_scores[i] = keep ? vals[src * AMDR_FUSE_NVALS + AMDR_FV_SCORE] : 0.0;
  out_mask[i] = keep ? mask[src] : 0;
  if (j == 0) out_count[q] = c;
}
}  // namespace amdr

extern "C" {

int amdr_fuse_device(const amdr_fuse_params_t* p, int32_t nq, const int64_t* dense_ids, const float* dense_scores,
                     int32_t kd, const int64_t* dense_row2uid, const int64_t* bm25_ids, const double* bm25_scores,
                     int32_t kb, const int64_t* bm25_row2uid, const int64_t* colbert_ids, const float* colbert_scores,
                     int32_t kc, const int64_t* colbert_row2uid, int64_t* out_ids, double* out_vals,
                     int32_t* out_mask, int32_t* out_count, int32_t device, void* stream) {
  int rc = fuse_check(p, nq, kd, kb, kc);
  if (rc) return rc;
  AMDR_REQUIRE((kd == 0 || (dense_ids && dense_scores)) && (kb == 0 || (bm25_ids && bm25_scores)) &&
                   (kc == 0 || (colbert_ids && colbert_scores)),
               "fuse: null channel buffer");
  AMDR_REQUIRE(nq == 0 || (out_ids && out_vals && out_mask && out_count), "fuse: null output");
  if (nq == 0) return AMDR_OK;
  AMDR_HIP(hipSetDevice(device));
  ChanIn c0{(const long long*)dense_ids, dense_scores, (const long long*)dense_row2uid, kd, 0};
  ChanIn c1{(const long long*)bm25_ids, bm25_scores, (const long long*)bm25_row2uid, kb, 1};
  ChanIn c2{(const long long*)colbert_ids, colbert_scores, (const long long*)colbert_row2uid, kc, 0};
  launch_fuse(*p, c0, c1, c2, nq, kd + kb + kc, (long long*)out_ids, out_vals, out_mask, out_count, (hipStream_t)stream);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int amdr_rerank_blend_device(int32_t nq, int32_t max_out, const int32_t* count, int64_t* ids, double* vals,
                             int32_t* mask, const double* ce_raw, int32_t top_n, double beta, double* out_rerank,
                             int32_t device, void* stream) {
  AMDR_REQUIRE(nq >= 0 && max_out >= 1 && max_out <= kFuseMax, "rerank_blend: bad sizes");
  AMDR_REQUIRE(top_n >= 1, "rerank_blend: top_n=%d", top_n);
  AMDR_REQUIRE(nq == 0 || (count && ids && vals && mask && ce_raw && out_rerank), "rerank_blend: null buffer");
  if (nq == 0) return AMDR_OK;
  AMDR_HIP(hipSetDevice(device));
  size_t lds = rerank_lds(max_out);
  if (lds > 48 * 1024)
    AMDR_HIP(hipFuncSetAttribute((const void*)rerank_blend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(rerank_blend_kernel, dim3(nq), dim3(64), lds, (hipStream_t)stream, max_out, count,
                     (long long*)ids, vals, mask, ce_raw, top_n, beta, out_rerank);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int amdr_fuse_compact_device(int32_t nq, int32_t max_out, int32_t w, const int64_t* ids, const double* vals,
                             const int32_t* mask, const int32_t* count, int64_t* out_rows, double* out_scores,
                             int32_t* out_mask, int32_t* out_count, int32_t device, void* stream) {
  AMDR_REQUIRE(nq >= 0 && max_out >= 1 && w >= 1 && w <= max_out, "fuse_compact: bad sizes (max_out=%d w=%d)", max_out, w);
  if (nq == 0) return AMDR_OK;
  AMDR_REQUIRE(ids && vals && mask && count && out_rows && out_scores && out_mask && out_count, "fuse_compact: null buffer");
  AMDR_HIP(hipSetDevice(device));
  const long total = (long)nq * w;
  hipLaunchKernelGGL(amdr::fuse_compact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)ids, vals, mask, count, nq, max_out, w, (long long*)out_rows, out_scores, out_mask,
                     out_count);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int amdr_fuse(const amdr_fuse_params_t* p, int32_t nq, const int64_t* dense_ids, const double* dense_scores, int32_t kd,
              const int64_t* bm25_ids, const double* bm25_scores, int32_t kb, const int64_t* colbert_ids,
              const double* colbert_scores, int32_t kc, int64_t* out_ids, double* out_vals, int32_t* out_mask,
              int32_t* out_count) {
  int rc = fuse_check(p, nq, kd, kb, kc);
  if (rc) return rc;
  if (nq == 0) return AMDR_OK;
  AMDR_REQUIRE((kd == 0 || (dense_ids && dense_scores)) && (kb == 0 || (bm25_ids && bm25_scores)) &&
                   (kc == 0 || (colbert_ids && colbert_scores)) && out_ids && out_vals && out_mask && out_count,
               "fuse: null buffer");
  int dev = 0;
  AMDR_HIP(hipGetDevice(&dev));
  const size_t q = (size_t)nq, mo = (size_t)(kd + kb + kc);
  Pack in(host_block());
  const size_t o_di = in.add(dense_ids, q * kd * 8), o_ds = in.add(dense_scores, q * kd * 8);
  const size_t o_bi = in.add(bm25_ids, q * kb * 8), o_bs = in.add(bm25_scores, q * kb * 8);
  const size_t o_ci = in.add(colbert_ids, q * kc * 8), o_cs = in.add(colbert_scores, q * kc * 8);
  const size_t in_bytes = in.used;
  // outputs follow the inputs in the same device arena, contiguous so one copy brings them back
  const size_t o_oi = in_bytes, o_ov = o_oi + pad256(q * mo * 8), o_om = o_ov + pad256(q * mo * AMDR_FUSE_NVALS * 8),
               o_oc = o_om + pad256(q * mo * 4), total = o_oc + pad256(q * 4);
  if ((rc = arena().begin(dev, total))) return rc;
  char* d = arena().base;
  hipStream_t st = nullptr;
  // the pageable block is sized for the outputs BEFORE the asynchronous copy reads it: growing
  // it afterwards would free the source of a copy that is nominally still in flight
  std::vector<char>& hb = host_block();
  if (hb.size() < total) hb.resize(total);
  AMDR_HIP(hipMemcpyAsync(d, hb.data(), in_bytes, hipMemcpyHostToDevice, st));
  ChanIn c0{(const long long*)(d + o_di), d + o_ds, nullptr, kd, 1};
  ChanIn c1{(const long long*)(d + o_bi), d + o_bs, nullptr, kb, 1};
  ChanIn c2{(const long long*)(d + o_ci), d + o_cs, nullptr, kc, 1};
  launch_fuse(*p, c0, c1, c2, nq, (int)mo, (long long*)(d + o_oi), (double*)(d + o_ov), (int*)(d + o_om),
              (int*)(d + o_oc), st);
  AMDR_HIP(hipGetLastError());
  AMDR_HIP(hipMemcpyAsync(hb.data() + o_oi, d + o_oi, total - o_oi, hipMemcpyDeviceToHost, st));
  AMDR_HIP(hipStreamSynchronize(st));
  memcpy(out_ids, hb.data() + o_oi, q * mo * 8);
  memcpy(out_vals, hb.data() + o_ov, q * mo * AMDR_FUSE_NVALS * 8);
  memcpy(out_mask, hb.data() + o_om, q * mo * 4);
  memcpy(out_count, hb.data() + o_oc, q * 4);
  return AMDR_OK;
}

int amdr_rerank_blend(int32_t nq, int32_t max_out, const int32_t* count, int64_t* ids, double* vals, int32_t* mask,
                      const double* ce_raw, int32_t top_n, double beta, double* out_rerank) {
  AMDR_REQUIRE(nq >= 0 && max_out >= 1 && max_out <= kFuseMax && top_n >= 1, "rerank_blend: bad sizes");
  if (nq == 0) return AMDR_OK;
  AMDR_REQUIRE(count && ids && vals && mask && ce_raw && out_rerank, "rerank_blend: null buffer");
  int dev = 0, rc;
  AMDR_HIP(hipGetDevice(&dev));
  const size_t q = (size_t)nq, mo = (size_t)max_out;
  Pack in(host_block());
  const size_t o_c = in.add(count, q * 4), o_r = in.add(ce_raw, q * top_n * 8);
  // in/out arrays next, contiguous, so one copy each way covers ids, vals, mask (+ rerank out)
  const size_t o_i = in.add(ids, q * mo * 8), o_v = in.add(vals, q * mo * AMDR_FUSE_NVALS * 8),
               o_m = in.add(mask, q * mo * 4);
  const size_t in_bytes = in.used;
  const size_t o_o = in_bytes, total = o_o + pad256(q * mo * 16);
  if ((rc = arena().begin(dev, total))) return rc;
  char* d = arena().base;
  hipStream_t st = nullptr;
  // the pageable block is sized for the outputs BEFORE the asynchronous copy reads it: growing
  // it afterwards would free the source of a copy that is nominally still in flight
  std::vector<char>& hb = host_block();
  if (hb.size() < total) hb.resize(total);
  AMDR_HIP(hipMemcpyAsync(d, hb.data(), in_bytes, hipMemcpyHostToDevice, st));
  rc = amdr_rerank_blend_device(nq, max_out, (const int32_t*)(d + o_c), (int64_t*)(d + o_i), (double*)(d + o_v),
                                (int32_t*)(d + o_m), (const double*)(d + o_r), top_n, beta, (double*)(d + o_o), dev, st);
  if (rc) return rc;
  AMDR_HIP(hipMemcpyAsync(hb.data() + o_i, d + o_i, total - o_i, hipMemcpyDeviceToHost, st));
  AMDR_HIP(hipStreamSynchronize(st));
  memcpy(ids, hb.data() + o_i, q * mo * 8);
  memcpy(vals, hb.data() + o_v, q * mo * AMDR_FUSE_NVALS * 8);
  memcpy(mask, hb.data() + o_m, q * mo * 4);
  memcpy(out_rerank, hb.data() + o_o, q * mo * 16);
  return AMDR_OK;
}

int amdr_hybrid_small_device(amdr_dense_t* dense, amdr_bm25_t* bm25, const float* Q_dev, const int32_t* q_terms_dev,
                             const int64_t* q_ptr_dev, int32_t nq, int32_t kd, int32_t kb, const amdr_fuse_params_t* p,
                             const int64_t* dense_row2uid, const int64_t* bm25_row2uid, float* dense_scores_dev,
                             int64_t* dense_ids_dev, double* bm25_scores_dev, int64_t* bm25_ids_dev, int64_t* out_ids,
                             double* out_vals, int32_t* out_mask, int32_t* out_count, void* stream) {
  AMDR_REQUIRE(dense && bm25, "hybrid_small: null index handle");
  AMDR_REQUIRE(p != nullptr, "hybrid_small: null params");
  AMDR_REQUIRE(p->method >= 0 && p->method <= AMDR_FUSE_WEIGHTED_SUM, "hybrid_small: method=%d", p->method);
  AMDR_REQUIRE(nq >= 0 && kd >= 1 && kd <= AMDR_MAX_K && kb >= 1 && kb <= AMDR_MAX_K, "hybrid_small: bad sizes");
  if (nq == 0) return AMDR_OK;
  AMDR_REQUIRE(Q_dev && q_terms_dev && q_ptr_dev, "hybrid_small: null query buffers");
  AMDR_REQUIRE(dense_scores_dev && dense_ids_dev && bm25_scores_dev && bm25_ids_dev, "hybrid_small: null channel lists");
  AMDR_REQUIRE(out_ids && out_vals && out_mask && out_count, "hybrid_small: null output");
  int64_t nd = 0, nb = 0;
  int rc;
  if ((rc = amdr_dense_ntotal(dense, &nd))) return rc;
  if ((rc = amdr_bm25_ndocs(bm25, &nb))) return rc;
  AMDR_REQUIRE(kd <= nd || nd == 0, "hybrid_small: kd=%d > %lld rows", kd, (long long)nd);
  bool one = false;
  if (nd >= 1 && nb >= 1 && nq <= 4 && dense_device_of(dense) >= 0) {
    std::lock_guard<std::mutex> gd(dense_mutex(dense));
    std::lock_guard<std::mutex> gb(bm25_mutex(bm25));
    AMDR_HIP(hipSetDevice(dense_device_of(dense)));
    Bm25Raw br;
    if ((rc = bm25_small_raw(bm25, nq, kb, &br))) return rc;
    if (hybrid_small_applies((long)nd, (long)nb, br.nslabs, nq, kd, kb)) {
      DenseRaw dr;
      if ((rc = dense_small_raw(dense, nq, &dr))) return rc;
      one = true;
      rc = hybrid_small_launch(dr, br, Q_dev, q_terms_dev, (const long long*)q_ptr_dev, nq, kd, kb, *p, dense_row2uid,
                               bm25_row2uid, dense_scores_dev, dense_ids_dev, bm25_scores_dev, bm25_ids_dev, out_ids,
                               out_vals, out_mask, out_count, (hipStream_t)stream);
    }
  }
  if (one) return rc;
  if ((rc = amdr_bm25_search_device(bm25, q_terms_dev, q_ptr_dev, nq, kb, bm25_scores_dev, bm25_ids_dev, stream))) return rc;
  return amdr_dense_search_fuse_device(dense, Q_dev, nq, kd, p, dense_row2uid, bm25_ids_dev, bm25_scores_dev, kb,
                                       bm25_row2uid, dense_scores_dev, dense_ids_dev, out_ids, out_vals, out_mask,
                                       out_count, stream);
}


}  // extern "C"
