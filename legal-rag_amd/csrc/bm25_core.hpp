// Device code of the BM25 channel (see bm25.hip for the design notes): scoring of one (query, slab) + ranking, shared by
// bm25.hip's kernels and the one-launch serving step in fuse.hip.  Both translation units are compiled with
// -ffp-contract=off (fp64, operand for operand what rank_bm25's numpy expression evaluates).
#pragma once
#include "common.hpp"
#include "topk.hpp"

#include <cfloat>
#include <cmath>

namespace amdr {

constexpr int kBmWaves = 4;
constexpr int kBmArgmaxK = 96;  // deepest k ever ranked by arg-max rounds (one round per result)
// Arg-max rounds cost ~ k x (scores per lane); the staged selector is nearly flat in k.  Measured
// crossover (scripts/sweep_bm25.py, 591 and 1 260 documents, k = 10 ... 80): k x ceil(slab / 64) ~ 480.
__host__ __device__ inline bool bm_use_argmax(int k, int slab) {
  return slab <= 2048 && (k <= 16 || (k <= kBmArgmaxK && k * ((slab + 63) >> 6) <= 480));
}
constexpr int kBmTok = 32;  // query tokens resolved per group (lanes fetch them in parallel)
constexpr int kBmOneWaveDocs = 2048;  // slabs up to this size are scored and ranked by ONE wave (<= 32 scores per lane)

__device__ __forceinline__ long uniform_i64(long v) {  // value known to be the same in every lane -> scalar pair
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v & 0xffffffffu));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}

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

// Top-k of a short score slab (m <= 64 * NV documents) by k rounds of a wave-wide arg-max on the
// fp64 scores held in registers, NV per lane (document lane + 64 v in register v).  One round:
//   lane maximum (NV-1 v_max_f64) -> wave maximum (6 exchange steps on DPP / permlane swaps)
//   -> per register one v_cmp_eq against it; the 64-bit lane masks are SCALARS, so "lowest
//   register with a hit, lowest lane in it" (ties -> lower doc id, the stable sort of
//   bm25_retriever.py:75) is scalar-unit work and needs no second cross-lane reduction
//   -> the winner's register is reset in its lane (two v_cndmask per register).
// ~6 NV + 25 vector instructions per round; the staged selector of topk.hpp sorts 128-bit
// candidates through LDS twice (measured 65 of the kernel's 98 us per 9 344 queries).
// -0.0 is folded to +0.0 and NaN ranks below every real score, as C64::make orders them.
// score as it is ranked: -0.0 -> +0.0, NaN below every real score
__device__ __forceinline__ double bm25_ranked(double raw) {
  const double x = raw + 0.0;
  return (x != x) ? -DBL_MAX : x;
}

template <int NV>
__device__ __forceinline__ int bm25_argmax_rounds(const double* sc, int m, int k, long lo, int lane, C64* out) {
  const double ninf = -INFINITY;
  double sv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) sv[v] = (lane + 64 * v < m) ? bm25_ranked(sc[lane + 64 * v]) : ninf;
  int got = 0;
#if defined(AMDR_BM_ABL) && AMDR_BM_ABL == 1
  for (int it = 0; it < 1; ++it) {
#else
  for (int it = 0; it < k; ++it) {
#endif
    double lm = sv[0];
#pragma unroll
    for (int v = 1; v < NV; ++v) lm = max_f64_raw(lm, sv[v]);
    const double wm = wave_allmax_f64(lm);
    if (wm == ninf) break;  // fewer than k documents in the slab
    unsigned long long fm = 0ull;
    int fv = 0;
#pragma unroll
    for (int v = NV - 1; v >= 0; --v) {
      const unsigned long long hit = __ballot(sv[v] == wm);
      fm = hit ? hit : fm;
      fv = hit ? v : fv;
    }
    const int wl = (int)__builtin_ctzll(fm);
    const bool mine = (lane == wl);
#pragma unroll
    for (int v = 0; v < NV; ++v) sv[v] = (mine && v == fv) ? ninf : sv[v];
    if (lane == 0) out[it] = C64::make(wm, lo + wl + 64 * fv);
    got = it + 1;
  }
  return got;
}

// The same ranking by a cheaper route, taken first: candidates are chosen on fp32 IMAGES of the
// fp64 scores (32-bit integer keys at the full vector rate instead of k rounds of half-rate fp64
// compares and a scalar mask search per register), then the exact fp64 order is CHECKED, not assumed.  Rounding is monotone (s1 > s2 => f(s1) >=
// f(s2)), so every document whose image is below the image of the k-th best cannot be in the
// top k; all documents whose image reaches the cut take part (the threshold test looks at the
// score half of the key only).  After the survivors are sorted by (image desc, doc asc) the
// order can differ from the exact (score desc, doc asc) order only inside a run of equal
// images: if any neighbouring pair up to the cut has equal images but different fp64 scores the
// function gives up (-1) and the caller runs the exact arg-max rounds.  More than 64 survivors
// (mass ties at the cut: e.g. every document at score 0) -> -1 as well.
template <int NV>
__device__ __forceinline__ int bm25_select_f32(const double* sc, int m, int k, long lo, int lane, C32* scratch,
                                               double* xs /* [64] exact scores of the survivors */, C64* out) {
  // fp32 images as order-preserving 32-bit keys (ord32: -0.0 -> +0.0, NaN lowest; 0 = no document)
  u32 img[NV];
  u32 lb = 0u;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int i = lane + 64 * v;
    img[v] = (i < m) ? ord32((float)sc[lane + 64 * v]) : 0u;
    lb = img[v] > lb ? img[v] : lb;
  }
  // The cut: the k-th largest of the 64 lane bests bounds the k-th best image from below.  It is
  // found from the top bit down with one compare + ballot per bit — the candidate and the count live
  // in scalar registers — instead of a 21-stage bitonic sort of 64-bit keys across the lanes (the
  // sort was ~300 of the kernel's ~1 300 vector instructions per wave).
  u32 T = 0u;
#pragma unroll
  for (int bit = 31; bit >= 0; --bit) {
    const u32 cand = T | (1u << bit);
    T = (__popcll(__ballot(lb >= cand)) >= k) ? cand : T;
  }
  // Survivors: key = image | ~document (16 bits: a slab has <= 2 048) | slot; the slot finds the
  // survivor's exact fp64 score (parked beside the key) again after the sort.
  // (each lane counts its own survivors, one inclusive prefix sum over the wave hands out the slots: ~5 vector
  // instructions per score register instead of a ballot / popcount / mbcnt round per register)
  const u32 Te = T > 1u ? T : 1u;
  int mine = 0;
#pragma unroll
  for (int v = 0; v < NV; ++v) mine += (img[v] >= Te) ? 1 : 0;
  int incl = mine;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int o = __shfl_up(incl, s);
    incl += (lane >= s) ? o : 0;
  }
  int cnt = __builtin_amdgcn_readlane(incl, 63);
  const bool overflow = cnt > 64;
  if (!overflow) {
    int at = incl - mine;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (img[v] >= Te) {
        C32 c;
        c.c = ((u64)img[v] << 32) | (u64)(((0xffffu - (u32)(lane + 64 * v)) << 8) | (u32)at);
        scratch[at] = c;
        xs[at] = bm25_ranked(sc[lane + 64 * v]);
        ++at;
      }
    }
  }
  if (overflow) {
    // More than 64 documents reach the cut: a mass tie AT the cut (the common case: every document the
    // query's tokens do not touch sits at score 0 — a third of the UCC-en evaluation queries have no token
    // the index knows).  Documents above the cut all rank first; if the documents AT the cut share one exact
    // fp64 score, the rest of the top k are simply the lowest ids among them.  Anything else -> undecided.
    int above = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) above += __popcll(__ballot(img[v] > T));
    if (T == 0u || above > 64) return -1;
    int need = k - above;  // documents still to take from the tie, lowest ids first (v-major = id order)
    need = need < 0 ? 0 : need;
    if (need > 0) {
      double x0 = 0.0;
      bool have = false;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const u64 mk = __ballot(img[v] == T);
        if (!have && mk) {
          x0 = __shfl(bm25_ranked(sc[lane + 64 * v]), (int)__builtin_ctzll(mk));
          have = true;
        }
      }
      u64 differ = 0ull;
#pragma unroll
      for (int v = 0; v < NV; ++v) differ |= __ballot(img[v] == T && bm25_ranked(sc[lane + 64 * v]) != x0);
      if (differ) return -1;
    }
    cnt = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const bool hi = img[v] > T;
      const bool eq = img[v] == T;
      const u64 mh = __ballot(hi), me = __ballot(eq);
      const int before_h = (int)__builtin_amdgcn_mbcnt_hi((u32)(mh >> 32), __builtin_amdgcn_mbcnt_lo((u32)mh, 0u));
      const int before_e = (int)__builtin_amdgcn_mbcnt_hi((u32)(me >> 32), __builtin_amdgcn_mbcnt_lo((u32)me, 0u));
      const int nh = __popcll(mh);
      int ne = __popcll(me);
      ne = ne < need ? ne : need;
      const bool take = hi || (eq && before_e < ne);
      if (take) {
        const int at = cnt + (hi ? before_h : nh + before_e);
        C32 c;
        c.c = ((u64)img[v] << 32) | (u64)(((0xffffu - (u32)(lane + 64 * v)) << 8) | (u32)at);
        scratch[at] = c;
        xs[at] = bm25_ranked(sc[lane + 64 * v]);
      }
      cnt += nh + ne;
      need -= ne;
    }
  }
  wave_lds_fence();
  C32 c = (lane < cnt) ? scratch[lane] : C32::pad();
  if (cnt <= 16)
    c = wave_sortN_desc<C32, 16>(c, lane);
  else if (cnt <= 32)
    c = wave_sortN_desc<C32, 32>(c, lane);
  else
    c = wave_sortN_desc<C32, 64>(c, lane);
  // exact scores of the survivors, and the check of every neighbouring pair up to the cut
  const u32 low = (u32)c.c;
  const int doc = (lane < cnt) ? (int)(0xffffu - (low >> 8)) : 0;
  const double x = xs[(lane < cnt) ? (int)(low & 63u) : 0];
  const u32 im = (u32)(c.c >> 32);
  const u32 im_n = (u32)__shfl_down((int)im, 1);
  const double x_n = __shfl_down(x, 1);
  const bool undecided = lane < k && lane + 1 < cnt && im == im_n && x != x_n;
  if (__ballot(undecided)) return -1;
  const int got = cnt < k ? cnt : k;
  if (lane < got) out[lane] = C64::make(x, lo + doc);
  return got;
}

// A query none of whose tokens has a posting in the slab leaves every score at +0.0: the ranking is the
// slab's first k documents (ties -> lower id).  No scoring, no selection.
__device__ __forceinline__ void bm25_all_zero_result(int m, int k, long lo, int lane, size_t row, double* fin_scores,
                                                     long long* fin_ids, C64* part_row) {
  for (int j = lane; j < k; j += 64) {
    const bool v = j < m;
    if (fin_ids) {
      fin_scores[row * k + j] = v ? 0.0 : -DBL_MAX;
      fin_ids[row * k + j] = v ? lo + j : -1ll;
    } else {
      part_row[j] = v ? C64::make(0.0, lo + j) : C64::pad();
    }
  }
}

// Ranking of a one-wave slab: candidates on fp32 images first
// (k <= 64), exact arg-max rounds when that is undecided or switched off.  Returns the count.
template <int NV>
__device__ __forceinline__ int bm25_rank_slab(const double* sc, int m, int k, long lo, int lane, bool use_select,
                                              C32* scratch, double* xs, C64* out) {
  int got = -1;
  if (k <= 64 && use_select) {
    got = bm25_select_f32<NV>(sc, m, k, lo, lane, scratch, xs, out);
    wave_lds_fence();
  }
  if (got < 0) {
    got = bm25_argmax_rounds<NV>(sc, m, k, lo, lane, out);
    wave_lds_fence();
  }
  return got;
}

// grid: (x = doc slabs, y = queries).  LDS: double sc[slab] + C64 lists[WAVES][cap] + int cnts[4] + token table [64]
// The host launches WAVES = 1 only (one wave per (query, slab): no block barriers, no list
// combine) — the 4-wave form the template still allows lost at every corpus size measured
// (bm_plan).  With a single slab the final (scores, ids) are written directly and the merge
// launch is skipped.
// NVT: scores per lane the register ranking is compiled for (>= ceil(slab / 64); the host picks the bucket — one
// kernel with every bucket inside carried the 32-register variant's VGPR count, 105, at every corpus size).
// One (query, slab) of the BM25 channel by the WAVES waves that call it (threadIdx.x < WAVES * 64; WAVES = 1: a single
// wave, no block barrier inside) — the body of bm25_score_topk_kernel, callable from the one-launch serving step of
// fuse.hip (hybrid_small_kernel) as well: the same instructions, hence the same bits.
template <int WAVES, int NVT>
__device__ __forceinline__ void bm25_block_query(
    const long long* __restrict__ term_ptr, const int* __restrict__ post_doc, const double* __restrict__ post_w,
    const double* __restrict__ idf, long n_terms, long n_docs, const int* __restrict__ q_terms,
    const long long* __restrict__ q_ptr, int nq, int k, int cap, int slab, int use_select,
    double* __restrict__ scores_out /* nullable [nq, n_docs] */, C64* __restrict__ part /* nullable [nslabs][nq][k] */,
    double* __restrict__ fin_scores /* nullable [nq,k]: single slab */, long long* __restrict__ fin_ids,
    int qi_arg, int slab_ix, unsigned char* smem) {
  double* sc = reinterpret_cast<double*>(smem);
  C64* lists = reinterpret_cast<C64*>(sc + slab);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)WAVES * cap);
  // Token table: posting range + idf of up to kBmTok query tokens at a time (32: a UCC-en query has 19 tokens; the
  // 1 KiB region is reused by the ranking — 64 survivor keys + 64 exact scores — and a 64-token table made it
  // 1.5 KiB: 25 -> 27 resident waves per CU at UCC-en size).
  long* tk_ps = reinterpret_cast<long*>(cnts + 4);
  long* tk_pe = tk_ps + kBmTok;
  double* tk_w = reinterpret_cast<double*>(tk_pe + kBmTok);
  int* tk_n = reinterpret_cast<int*>(tk_ps + 128);  // tokens of the current group that have postings in this slab
  constexpr int NT = WAVES * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = qi_arg;
  const long lo = (long)slab_ix * slab;
  long hi = lo + slab;
  if (hi > n_docs) hi = n_docs;
  const int m = (int)(hi - lo);

  for (int i = tid; i < m; i += NT) sc[i] = 0.0;
  block_sync<WAVES>();

  // Token metadata (posting range inside this slab, idf) is fetched by the lanes IN PARALLEL,
  // kBmTok tokens at a time, and parked in LDS: walking the tokens one by one would chain three
  // dependent global loads (term id -> term_ptr -> postings) per token, ~1.5 us each.
  int nt_total = 0;  // block-uniform
  const long t0 = q_ptr[qi], t1 = q_ptr[qi + 1];
  for (long tb = t0; tb < t1; tb += kBmTok) {
    const int nt_all = (int)((t1 - tb) < kBmTok ? (t1 - tb) : kBmTok);
    if (tid < 64) {
      long ps = 0, pe = 0;
      double w = 0.0;
      if (tid < nt_all) {
        const int term = q_terms[tb + tid];
        if (term >= 0 && term < n_terms) {  // unknown token: idf 0, contributes +0.0 -> skipped
          const long p0 = term_ptr[term], p1 = term_ptr[term + 1];
          ps = (lo == 0) ? p0 : lower_bound_i32(post_doc, p0, p1, (int)lo);
          pe = (hi >= n_docs) ? p1 : lower_bound_i32(post_doc, ps, p1, (int)hi);
          w = idf[term];
        }
      }
      // Only tokens with postings in this slab are kept, in query order: a token without any adds
      // nothing to any score (jieba's query tokens include every blank and punctuation mark —
      // 13 of the 19 tokens of an average UCC-en query are unknown to the index).
      const bool keep = pe > ps;
      const unsigned long long km = __ballot(keep);
      const unsigned long long below = (tid == 0) ? 0ull : (km & (~0ull >> (64 - tid)));
      if (keep) {
        const int at = __popcll(below);
        tk_ps[at] = ps;
        tk_pe[at] = pe;
        tk_w[at] = w;
      }
      if (tid == 0) *tk_n = __popcll(km);
    }
    block_sync<WAVES>();
    const int nt = __builtin_amdgcn_readfirstlane(*tk_n);
    nt_total += nt;
    if (nt == 0) {
      block_sync<WAVES>();  // the table is rewritten by the next group
      continue;
    }
    // rank_bm25: idf * (q_freq * (k1 + 1) / (q_freq + k1 * (1 - b + b * doc_len / avgdl))); the
    // parenthesis depends only on (tf, doc) and was evaluated once at index creation.
    // Tokens are applied in query order (the accumulation order of rank_bm25); inside one token a
    // list holds a document once, so the scatter has no conflicts and needs no ordering.  The
    // list bounds are wave-uniform (scalar registers): the loops and the short-list case branch
    // on scalars, a lane's offset is 32-bit, and out-of-range lanes read a clamped address and
    // are masked at the update — no divergent branches around the loads.
    // The posting lists of the kept tokens are walked as ONE sequence of 64-posting chunks, eight
    // chunks (across token boundaries) requested before the first of them is applied: a wave then
    // waits for an L2 round trip once per 512 postings instead of once or twice per token (the
    // kernel spent 46 % of its wave cycles parked on s_waitcnt with the per-token prefetch).
    // Chunks are applied in sequence order — token order, the accumulation order of rank_bm25 — and
    // the LDS unit serves a wave's operations in order.
    constexpr int kAhead = 8;
    int t = 0, base = 0;
    long ps = uniform_i64(tk_ps[0]);
    int len = __builtin_amdgcn_readfirstlane((int)(tk_pe[0] - tk_ps[0]));
    double w = tk_w[0];
#if defined(AMDR_BM_ABL) && AMDR_BM_ABL == 2
    t = nt;
#endif
    while (t < nt) {
      int dd[kAhead];
      double ww[kAhead], wv[kAhead];
      bool ok[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const bool have = t < nt;  // wave-uniform
        const int j = base + tid;
        ok[u] = have && j < len;
        int jj = ok[u] ? j : len - 1;
        jj = jj < 0 ? 0 : jj;
        dd[u] = post_doc[ps + jj];  // unconditional, clamped: countable in s_waitcnt
        ww[u] = post_w[ps + jj];
        wv[u] = w;
        if (have) {
          base += NT;
          if (base >= len) {
            ++t;
            base = 0;
            if (t < nt) {
              ps = uniform_i64(tk_ps[t]);
              len = __builtin_amdgcn_readfirstlane((int)(tk_pe[t] - tk_ps[t]));
              w = tk_w[t];
            }
          }
        }
      }
      // One wave: the update is an LDS fp64 atomic add without return (ds_add_f64: the same correctly rounded
      // add, no read-back to wait for); the LDS unit serves a wave's operations in order, so chunk u + 1 — possibly
      // the next token hitting the same document — lands after chunk u.  (As ds_read / v_add_f64 / ds_write with a
      // fence per chunk the scatter was a chain of LDS round trips.)
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        if (WAVES == 1) {
          if (ok[u])
            __hip_atomic_fetch_add(&sc[dd[u] - lo], wv[u] * ww[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
          if (ok[u]) sc[dd[u] - lo] += wv[u] * ww[u];
          block_sync<WAVES>();
        }
      }
      if (WAVES == 1) wave_lds_fence();
    }
  }

  if (scores_out) {
    for (int i = tid; i < m; i += NT) scores_out[(size_t)qi * n_docs + lo + i] = sc[i];
  }
  if (!part && !fin_ids) return;
  if (nt_total == 0) {  // no token of the query has a posting here
    if (wave == 0)
      bm25_all_zero_result(m, k, lo, lane, (size_t)qi, fin_scores, fin_ids,
                           part ? part + ((size_t)slab_ix * nq + qi) * k : nullptr);
    return;
  }
#if defined(AMDR_BM_ABL) && AMDR_BM_ABL == 3  // timing-only build: no ranking
  if (fin_ids && tid < k) {
    fin_ids[(size_t)qi * k + tid] = tid;
    fin_scores[(size_t)qi * k + tid] = sc[tid];
  }
  return;
#endif

  WaveTopK<C64> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  // Short slab and shallow k (bm_use_argmax; the serving shape is 591 docs, k = 10): bm25_argmax_rounds.
  bool done = false;
  if (WAVES == 1 && bm_use_argmax(k, slab)) {
    C32* scratch = reinterpret_cast<C32*>(tk_ps);  // 64 x 8 B each: the token table is dead once the slab is scored
    double* xs = reinterpret_cast<double*>(tk_ps + 64);
    const int got = bm25_rank_slab<NVT>(sc, m, k, lo, lane, use_select != 0, scratch, xs, tk.buf);
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
      C64* dst = part + ((size_t)slab_ix * nq + qi) * k;
      for (int j = lane; j < k; j += 64) dst[j] = (j < tk.cnt) ? tk.buf[j] : C64::pad();
    }
  }
}

template <int WAVES, int NVT>
__global__ __launch_bounds__(WAVES * 64) void bm25_score_topk_kernel(
    const long long* __restrict__ term_ptr, const int* __restrict__ post_doc, const double* __restrict__ post_w,
    const double* __restrict__ idf, long n_terms, long n_docs, const int* __restrict__ q_terms,
    const long long* __restrict__ q_ptr, int nq, int k, int cap, int slab, int use_select,
    double* __restrict__ scores_out /* nullable [nq, n_docs] */, C64* __restrict__ part /* nullable [nslabs][nq][k] */,
    double* __restrict__ fin_scores /* nullable [nq,k]: single slab */, long long* __restrict__ fin_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bm25_block_query<WAVES, NVT>(term_ptr, post_doc, post_w, idf, n_terms, n_docs, q_terms, q_ptr, nq, k, cap, slab, use_select, scores_out, part, fin_scores, fin_ids,
                               (int)blockIdx.y, (int)blockIdx.x, smem);
}

static __global__ __launch_bounds__(256) void bm25_merge_kernel(const C64* __restrict__ part, int nparts, int nq, int k,
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
