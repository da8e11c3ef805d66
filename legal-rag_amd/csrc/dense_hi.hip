// Dense channel, large scans: the FIRST pass of the two-level top-k on the fp16 matrix instructions.
//
// The two-level top-k of dense.hip (run_search_two_level) scans the chunk matrix once for per-tile maxima, picks
// the candidate tiles of every query and re-scores those exactly.  With 32 queries per scan the exact-fp32 MFMAs of
// the first pass alone need 4.6 ms on the 10 M x 768 matrix, HBM alone 4.45 ms: the pass runs at 0.73 of the HBM
// peak and cannot hold more than 32 queries (a 96-KiB fp32 query tile in LDS).  The first pass only has to find
// candidate tiles, and the second pass is exact — so here it runs on v_mfma_f32_32x32x16_f16 with fp16 ROUNDINGS of
// both operands (hi parts, csrc/maxsim.hip): 16x the matrix rate, a 96-KiB tile now holds 64 queries, the scan is
// bound by HBM alone.  Every approximate tile maximum lies within a proven eps of the exact one (dense_hi_check_kernel
// states the bound), the candidate cut is widened by it (dense.hip run_search_two_level / two_level_pass), the
// re-scoring pass is the unchanged exact fp32 kernel: ids and score bits are those of the exact forms.  A query whose
// cut the bound does not separate (mass near-ties) raises a device flag and the exact first pass runs for that batch
// (gated launches: no host round trip).
//
// Kernel: a persistent block of 8 waves per CU; the query tile — up to 64 queries converted to fp16 (per-query
// power-of-two scale) — sits in LDS for the whole launch; every wave streams its own contiguous run of 32-row tiles of X:
// coalesced 16-B/lane fp32 loads (4 rows x 256 B per wave instruction, non-temporal, 2 chunks of 64 floats per row in
// flight across tile boundaries), conversion to fp16 in registers (the matrix's power-of-two scale applied), a
// wave-private 4-KiB fp16 stage in LDS, fragments back, 8 MFMAs per 64-float chunk (32 rows x 64 queries), one maximum per
// tile and query.  Two launches per search: a SAMPLE of every s-th tile writes its maxima (MT[item][queries]) and gives
// each query a threshold (the kc-th best sampled maximum, a lower bound of the kc-th best overall); the full SCAN then
// lets only the maxima that reach the threshold leave the kernel, as entries of one flat candidate list.  Writing all
// 20 M maxima of a 10 M-row scan cost 0.3-0.9 ms of 5 whatever the layout.
#include "common.hpp"
#include "topk.hpp"

#include <cfloat>
#include <cmath>
#include <cstdlib>

namespace amdr {

typedef float hi4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int kHiWaves = 8;
constexpr int kHiKC = 64;              // floats of every row per chunk
constexpr int kHiStageBytes = 32 * 128;  // 32 rows x 64 halves
constexpr int kHiWbufMax = 384;          // entries of a wave's staging buffer at most (hi_wbuf_entries)

// stage: row r (0..31) at byte r*128, its 16-B slot s (0..7) at s ^ ((r >> 1) & 7) (the image of dense_mfma.hip's stage)
__device__ __forceinline__ int hi_stage_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
// query tile: row q (0..63) at byte q * (d * 2), its 16-B chunk c at c ^ (q & 15) (rows alias on the banks: d * 2 % 256 == 0)
__device__ __forceinline__ int hi_q_off(int q, int chunk, int d) { return q * (d * 2) + ((chunk ^ (q & 15)) << 4); }
__host__ __device__ constexpr int hi_query_tile(int d) { return d > 896 ? 48 : 64; }

// grid: 1-D over row slabs; LDS: 64 x d halves (query tile) + kHiWaves x 4 KiB (stages).  M[q][tile] is the maximum of
// (x * x_scale) . (q * 2^-e_q) over the tile's rows — a per-query positive scaling of the scores, which is all a
// per-query ranking of tiles needs; keeping the scaled value keeps it clear of fp32's subnormal range
// EMIT = false: item i of the launch is tile i * tile_stride (a strided SAMPLE of the tiles), its maxima go to MT[i][cols].
// EMIT = true (the full scan, tile_stride 1): a tile's maximum leaves the kernel only if it reaches tau[query] — the kc-th
// best maximum of the sample, a lower bound of the kc-th best overall, so every tile of a query's global top-kc passes
// — as a packed (score, query << 26 | tile) entry of ONE flat list: staged in a wave-private LDS buffer, appended with one
// atomic per flush (normally one per wave and launch).  ~kc x tile_stride entries per query instead of every maximum.
struct HiEmit {
  const float* tau;     // [nq], stride tau_stride floats
  int tau_stride;
  C32* cand;            // flat list
  unsigned int* total;  // entries appended (may pass cap: the check kernel raises the flag)
  unsigned int cap;
  int wbuf;             // entries of the wave-private staging buffer
  // per-query lists (round 4; qcount != nullptr): query q's entries (score, tile) go to qlist[q * qcap ..], qcount[q] counts
  // them (it may pass qcap: dense_hi_select_kernel then raises the flag for that query).  The block bins the staged
  // entries of its 8 waves by query in LDS and reserves its share of every list with ONE atomic per query (256 blocks x
  // 64 atomics per scan), instead of one flat list that a second kernel had to filter per query (49 + 11 us).
  C32* qlist;
  unsigned int* qcount;
  unsigned int qcap;
  int n_qtiles;  // > 1: the launch scans the matrix once per query tile (tau / qlist / qcount of tile y at y x the tile size)
};
constexpr int kHiQShift = 26;  // tiles < 2^26

template <int D64, bool EMIT>  // d / 64
__device__ __forceinline__ void hi_tilemax_pass(const float* __restrict__ X, long n, const float* __restrict__ Q, int nq,
                                                float* __restrict__ MT /*[items][cols]*/, int cols, float x_scale,
                                                long tile_stride, long n_items, const HiEmit& em, unsigned char* smem,
                                                int qoff /* first query of this pass in tau / qlist / qcount */) {
  constexpr int d = D64 * 64;
  constexpr int NCH = D64;
  constexpr int QT = hi_query_tile(d);  // queries of the LDS tile: 64, or 48 at d = 1 024 (96 KiB either way at the widest)
  unsigned char* qt = smem;
  unsigned char* stage = smem + QT * d * 2 + (threadIdx.x >> 6) * kHiStageBytes;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;

  // grid.y > 1 (the sample of a multi-tile search): block row y works for query tile y
  // round 4 (sample only): em.tau_stride < 0 asks for the query-major layout MT[query][ld = -tau_stride] (one 4-byte store
  // per (item, query); the threshold kernel then reads a query's row with coalesced loads: 27 -> 6 us)
  const long q_ld = (!EMIT && em.tau_stride < 0) ? -(long)em.tau_stride : 0;
  if (gridDim.y > 1) {
    Q += (size_t)blockIdx.y * QT * d;
    nq -= (int)blockIdx.y * QT;
    nq = nq < 0 ? 0 : (nq > QT ? QT : nq);
    MT += q_ld ? (size_t)blockIdx.y * QT * q_ld : (size_t)blockIdx.y * n_items * cols;
  }
  // The wave's first two chunks of X are requested BEFORE the query tile is converted (they do not depend on it): the
  // two memory round trips at the head of every launch overlap instead of following each other.
  // Persistent block (one per CU: the query tile is converted once).  Every wave of the grid owns one contiguous run
  // of tiles (run lengths differ by at most one tile) and streams it front to back; its loads run two 64-float chunks
  // ahead, across tile boundaries.
  const long gw = (long)blockIdx.x * kHiWaves + wave, nw = (long)gridDim.x * kHiWaves;
  const long t_lo = n_items * gw / nw, t_hi = n_items * (gw + 1) / nw;  // this wave's run of items
  // loader role inside a 1-KiB piece: 4 rows x 256 B; lane: row l >> 4, 16-B piece l & 15 (4 floats)
  const int lrow = lane >> 4, lpiece = lane & 15;
  auto row_ptr = [&](long item, int p) {
    long r = item * tile_stride * 32 + 4 * p + lrow;
    if (r >= n) r = n - 1;  // rows past the end repeat the last row: no effect on a maximum
    return X + (size_t)r * d + lpiece * 4;
  };
  hi4f G[2][8];
  const float* gpn[8];
  if (t_lo < t_hi) {
#pragma unroll
    for (int p = 0; p < 8; ++p) gpn[p] = row_ptr(t_lo, p);
    // issue order = the steady state's (chunk 0's loads, then chunk 1's): the loop header's vmcnt waits are the
    // minimum over this entry and the back edge, an interleaved order here would drain the queue at every tile start
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int p = 0; p < 8; ++p) G[c][p] = __builtin_nontemporal_load(reinterpret_cast<const hi4f*>(gpn[p] + c * kHiKC));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the query tile: wave w converts queries 8 w .. 8 w + 7 (scale = 2^-e of the query's largest |component|).
  // Lane l owns the 16-byte fp16 chunks l and l + 64 of a query (8 consecutive components each: two 16-byte loads, one
  // 16-byte LDS store); the loads of FOUR queries are in flight before the first is reduced — one query at a time with
  // 4-byte loads, shuffles through LDS and 2-byte LDS stores was ~2 us per query, 16 us of every launch of this kernel.
  {
    constexpr int CPQ = d / 8, ITS = (CPQ + 63) / 64;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      hi4f v[4][ITS][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qi = wave * 8 + g * 4 + j;
        const bool live = qi < nq;
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int c = lane + 64 * it;
          const bool ok = live && c < CPQ;
          const float* src = Q + (size_t)(ok ? qi : 0) * d + (ok ? c : 0) * 8;
          const hi4f z = {0.f, 0.f, 0.f, 0.f};
          v[j][it][0] = ok ? *reinterpret_cast<const hi4f*>(src) : z;
          v[j][it][1] = ok ? *reinterpret_cast<const hi4f*>(src + 4) : z;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qi = wave * 8 + g * 4 + j;
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < ITS; ++it)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[j][it][hh][e]));
        // (fmaxf drops a NaN operand: a NaN / infinite query is caught by the check on the fp32 query itself)
        m = fmaxf(m, __uint_as_float(lane_xor<1>(__float_as_uint(m))));
        m = fmaxf(m, __uint_as_float(lane_xor<2>(__float_as_uint(m))));
        m = fmaxf(m, __uint_as_float(lane_xor<4>(__float_as_uint(m))));
        m = fmaxf(m, __uint_as_float(lane_xor<8>(__float_as_uint(m))));
        m = fmaxf(m, __uint_as_float(lane_xor<16>(__float_as_uint(m))));
        m = fmaxf(m, __uint_as_float(lane_xor<32>(__float_as_uint(m))));
        int e = 0;
        if (m > 0.f && m <= FLT_MAX) (void)frexpf(m, &e);
        const float sc = ldexpf(1.f, -e);
        if (qi < QT) {
#pragma unroll
          for (int it = 0; it < ITS; ++it) {
            const int c = lane + 64 * it;
            if (c < CPQ) {
              h8 y;
#pragma unroll
              for (int e2 = 0; e2 < 4; ++e2) {
                y[e2] = (_Float16)(v[j][it][0][e2] * sc);
                y[4 + e2] = (_Float16)(v[j][it][1][e2] * sc);
              }
              *reinterpret_cast<h8*>(qt + hi_q_off(qi, c, d)) = y;
            }
          }
        }
      }
    }
  }
  __syncthreads();

  float mt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) mt[j] = 0.f;
  // EMIT: the wave's staging buffer and its fill (wave-uniform), the lane's threshold
  C32* wbuf = reinterpret_cast<C32*>(smem + QT * d * 2 + kHiWaves * kHiStageBytes) + (size_t)wave * (EMIT ? em.wbuf : 0);
  int wcnt = 0;
  const float tau = (EMIT && lane < nq) ? em.tau[(size_t)(qoff + lane) * em.tau_stride] : 0.f;
  unsigned int* const qcount = EMIT ? em.qcount + qoff : nullptr;
  C32* const qlist = EMIT ? em.qlist + (size_t)qoff * em.qcap : nullptr;
  const bool perq = EMIT && em.qcount != nullptr;
  auto flush = [&]() {
    if (perq) {  // a full buffer in the middle of a run (rare): one returning atomic per entry
      for (int i = lane; i < wcnt; i += 64) {
        C32 c = wbuf[i];
        const unsigned int id = 0xffffffffu - (unsigned int)c.c;
        const unsigned int q = id >> kHiQShift;
        c.c = (c.c & 0xffffffff00000000ull) | (u64)(0xffffffffu - (id & ((1u << kHiQShift) - 1u)));
        const unsigned int pos = atomicAdd(qcount + q, 1u);
        if (pos < em.qcap) qlist[(size_t)q * em.qcap + pos] = c;
      }
    } else {
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(em.total, (unsigned int)wcnt);
      base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
      for (int i = lane; i < wcnt; i += 64)
        if (base + (unsigned int)i < em.cap) em.cand[base + i] = wbuf[i];
    }
    wcnt = 0;
    wave_lds_fence();
  };
  for (long t = t_lo; t < t_hi; ++t) {
    const bool has_next = t + 1 < t_hi;
    const float* gp[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      gp[p] = gpn[p];
      gpn[p] = row_ptr(has_next ? t + 1 : t, p);
    }
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[b][j] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // chunk c: fp32 -> fp16 (4 floats -> 8 bytes per piece), into the stage: row 4 p + lrow, 8-byte half (lpiece & 1) of
      // 16-B slot lpiece >> 1
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const hi4f x = G[c & 1][p];
        h4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (_Float16)(x[e] * x_scale);
        *reinterpret_cast<h4*>(stage + hi_stage_off(4 * p + lrow, lpiece >> 1) + (lpiece & 1) * 8) = y;
      }
      // the refill is issued HERE, two chunks ahead of its use (hipcc otherwise sinks the loads towards their first use
      // and the wave waits a full memory latency per chunk: sched_barrier pins them)
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < NCH) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
          G[c & 1][p] = __builtin_nontemporal_load(reinterpret_cast<const hi4f*>(gp[p] + (c + 2) * kHiKC));
      } else {  // NCH is even: chunk c + 2 - NCH of the next tile lands in the buffer of its parity.  Unconditional (the
                // run's last tile re-reads its own first chunks): behind a branch hipcc drains vmcnt to 0 at every tile end
#pragma unroll
        for (int p = 0; p < 8; ++p)
          G[c & 1][p] = __builtin_nontemporal_load(reinterpret_cast<const hi4f*>(gpn[p] + (c + 2 - NCH) * kHiKC));
      }
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence();
      // fragments: A = row r32, chunks 2 s + h (s = 0..3) of the stage; B = query r32 (+ 32), chunks 8 c + 2 s + h of its row
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const h8 a = *reinterpret_cast<const h8*>(stage + hi_stage_off(r32, 2 * s + h));
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          // (a 48-query tile has no rows 48..63: the second block's upper columns re-read rows 32..47 and are never used)
          const int qrow = b == 0 ? r32 : 32 + (QT == 64 ? r32 : (r32 & 15));
          const h8 q = *reinterpret_cast<const h8*>(qt + hi_q_off(qrow, 8 * c + 2 * s + h, d));
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, q, acc[b], 0, 0, 0);
        }
      }
      wave_lds_fence();  // the next chunk overwrites the stage
    }
    // tile maximum per query: 16 rows in the lane's registers, the other 16 in lane ^ 32; lane l keeps query l's
    float mq[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float m = acc[b][0];
#pragma unroll
      for (int j = 1; j < 16; ++j) m = fmaxf(m, acc[b][j]);
      mq[b] = fmaxf(m, __uint_as_float(lane_xor<32>(__float_as_uint(m))));
    }
    // In the scaled units of the lane's query (see the check kernel).
    const float mine = h ? mq[1] : mq[0];
    if (EMIT) {
      // What the maxima cost when all of them were written (timing-only build without any store: 4.38 ms per scan):
      // M[query][tile], a 4-byte store per (tile, query): 5.1-5.3 ms — 64 DRAM pages per instruction; MT[tile][queries]
      // rows + a transposition: 4.84 ms + 70 us + a 95-us top-k over 20 M values.  Here: only maxima that reach the
      // sample's threshold, through LDS.
      const bool pass = lane < nq && mine >= tau;
      const unsigned long long pm = __ballot(pass);
      if (pm != 0ull) {  // wave-uniform
        const unsigned long long below = lane == 0 ? 0ull : (pm & (~0ull >> (64 - lane)));
        if (pass) wbuf[wcnt + __popcll(below)] = C32::make(mine, ((unsigned int)lane << kHiQShift) | (unsigned int)t);
        wcnt += __popcll(pm);
        if (wcnt + 64 > em.wbuf) flush();
      }
    } else if (q_ld) {
      if (lane < nq) MT[(size_t)lane * q_ld + t] = mine;
    } else {
      // MT[item][cols], cols = the queries rounded up to 16: one contiguous row per item, the maxima of 8 items collected
      // in registers and written as 8 back-to-back row stores.
      const int j8 = (int)((t - t_lo) & 7);
#pragma unroll
      for (int j = 0; j < 8; ++j) mt[j] = j8 == j ? mine : mt[j];
      if (j8 == 7 || !has_next) {  // wave-uniform
        float* dst = MT + (size_t)(t - j8) * cols + lane;
        if (lane < cols) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j <= j8) dst[(size_t)j * cols] = mt[j];
        }
      }
    }
  }
  if (EMIT && !perq && wcnt > 0) flush();
  if (perq) {  // block-uniform: every wave of the block gets here
    // the staged entries of the block's 8 waves, binned by query: rank inside the block by LDS atomics, one global atomic
    // per query with an entry reserves the block's share of that query's list
    int* bh = reinterpret_cast<int*>(smem + QT * d * 2 + kHiWaves * kHiStageBytes + (size_t)kHiWaves * em.wbuf * sizeof(C32));
    unsigned int* bb = reinterpret_cast<unsigned int*>(bh + 64);
    if (wave == 0) bh[lane] = 0;
    __syncthreads();
    constexpr int kJ = kHiWbufMax / 64;
    int pos[kJ];
    unsigned int qq[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      const int i = lane + 64 * j;
      pos[j] = 0, qq[j] = 0u;
      if (i < wcnt) {
        qq[j] = (0xffffffffu - (unsigned int)wbuf[i].c) >> kHiQShift;
        pos[j] = atomicAdd(bh + qq[j], 1);
      }
    }
    __syncthreads();
    if (wave == 0) {
      const int c = bh[lane];
      bb[lane] = c > 0 ? atomicAdd(qcount + lane, (unsigned int)c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      const int i = lane + 64 * j;
      if (i < wcnt) {
        C32 c = wbuf[i];
        const unsigned int id = 0xffffffffu - (unsigned int)c.c;
        c.c = (c.c & 0xffffffff00000000ull) | (u64)(0xffffffffu - (id & ((1u << kHiQShift) - 1u)));
        const unsigned int dst = bb[qq[j]] + (unsigned int)pos[j];
        if (dst < em.qcap) qlist[(size_t)qq[j] * em.qcap + dst] = c;
      }
    }
  }
}

// The kernel: one pass per query tile.  The scan of a search with several query tiles (round 4: up to four per tail) is ONE
// launch that walks the tiles — convert tile y, scan, flush, next tile — instead of one launch per tile: no drain and
// refill of the chip between the scans (10 us each) and the waves' uneven ends overlap with the next tile's start.
template <int D64, bool EMIT>
__global__ __launch_bounds__(kHiWaves * 64) void dense_hi_tilemax_kernel(const float* __restrict__ X, long n,
                                                                         const float* __restrict__ Q, int nq,
                                                                         float* __restrict__ MT /*[items][cols]*/,
                                                                         int cols, float x_scale, long tile_stride,
                                                                         long n_items, HiEmit em) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int QT = hi_query_tile(D64 * 64);
  const int tiles = (EMIT && em.n_qtiles > 1) ? em.n_qtiles : 1;
#pragma unroll 1
  for (int y = 0; y < tiles; ++y) {  // (one call site: a second inlined copy of the pass spilled registers)
    int nq_y = nq;  // (the SAMPLE of a multi-tile search runs one tile per block row: it takes its share of nq itself)
    if (tiles > 1) {
      nq_y = nq - y * QT;
      nq_y = nq_y > QT ? QT : nq_y;
    }
    hi_tilemax_pass<D64, EMIT>(X, n, Q + (size_t)y * QT * (D64 * 64), nq_y, MT, cols, x_scale, tile_stride, n_items, em, smem,
                               y * QT);
    if (y + 1 < tiles) __syncthreads();  // every wave is done with this tile's queries and lists before the next is converted
  }
}

// Top-kc of every query from the flat list of the emitting scan, step 1: block (part p, query q) reads the p-th slice of
// the list (from L2), keeps the query's entries, writes its best kc as part[p][q][kc] — the layout dense_merge_kernel
// merges.  (score desc, tile asc) like every other top-k here.  One block per query over the whole list took 227 us.
constexpr int kHiCandParts = 16;
__global__ __launch_bounds__(256) void dense_hi_cand_topk_kernel(const C32* __restrict__ cand,
                                                                 const unsigned int* __restrict__ total, unsigned int cap,
                                                                 int nq, int kc, int tcap, C32* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)4 * tcap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned int qi = blockIdx.y;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * tcap, tcap, kc);
  unsigned int n = *total;
  if (n > cap) n = cap;
  const unsigned int per = (n + gridDim.x - 1) / gridDim.x;
  const unsigned int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  // eight entries per lane are requested before the first is looked at: one load per trip was one L2 round trip per 256
  // entries and wave (66 trips, ~40 of the kernel's 49 us)
  constexpr int UN = 8;
  for (unsigned int base = lo + (unsigned int)wave * 64; base < hi; base += 256 * UN) {
    C32 e[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const unsigned int i = base + 256 * u + lane;
      e[u] = i < hi ? cand[i] : C32::pad();
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      C32 c = e[u];
      const unsigned int id = 0xffffffffu - (unsigned int)c.c;
      const bool v = !c.is_pad() && (id >> kHiQShift) == qi;
      c.c = (c.c & 0xffffffff00000000ull) | (u64)(0xffffffffu - (id & ((1u << kHiQShift) - 1u)));
      tk.push_lanes(c, v, lane);
    }
  }
  tk.finalize(lane);
  block_combine_topk(tk, lists, tcap, 4, wave, lane, cnts);
  if (wave == 0) {
    C32* out = part + ((size_t)blockIdx.x * nq + qi) * kc;
    for (int j = lane; j < kc; j += 64) out[j] = j < tk.cnt ? tk.buf[j] : C32::pad();
  }
}

// MT[tile][cols] -> M[q][tile] (the layout the top-k pass reads): 256 tiles per block through LDS, 1-KiB runs per query row
constexpr int kHiTrTiles = 256;
__host__ __device__ inline int hi_mt_cols(int nq) { return (nq + 15) & ~15; }
__global__ __launch_bounds__(256) void dense_hi_transpose_kernel(const float* __restrict__ MT, long tiles, int nq, long ldM,
                                                                 float* __restrict__ M, int tpb /* <= kHiTrTiles, x 64 */) {
  __shared__ float patch[kHiTrTiles][65];
  const long t0 = (long)blockIdx.x * tpb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cols = hi_mt_cols(nq);
  for (int r = wave; r < tpb; r += 4)
    if (t0 + r < tiles && lane < cols) patch[r][lane] = MT[(size_t)(t0 + r) * cols + lane];
  __syncthreads();
  for (int q = wave; q < nq; q += 4)
    for (int r = lane; r < tpb; r += 64)
      if (t0 + r < tiles) M[(size_t)q * ldM + t0 + r] = patch[r][q];
}

// largest |component| and largest row L2 norm of a matrix (float bits; both are non-negative)
__global__ __launch_bounds__(256) void dense_stats_kernel(const float* __restrict__ X, long n, int d,
                                                          unsigned int* __restrict__ out2) {
  const int lane = threadIdx.x & 63;
  const long wv = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (long)gridDim.x * 4;
  float amax = 0.f, nmax = 0.f;
  for (long r = wv; r < n; r += nw) {
    float ss = 0.f;
    for (int k = lane; k < d; k += 64) {
      const float x = X[(size_t)r * d + k];
      amax = fmaxf(amax, fabsf(x));
      ss += x * x;
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) ss += __shfl_xor(ss, sft);
    nmax = fmaxf(nmax, sqrtf(ss));
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) amax = fmaxf(amax, __shfl_xor(amax, sft));
  if (lane == 0) {
    atomicMax(out2, __float_as_uint(amax != amax ? INFINITY : amax));
    atomicMax(out2 + 1, __float_as_uint(nmax != nmax ? INFINITY : nmax));
  }
}

// After the top-kc1 of the approximate tile maxima (vals [m][kc1], descending, in each query's scaled units): is every
// tile that can hold one of a query's k best rows among its first kc1 - 1?  Yes if the kc1-th maximum lies below
// T_k - 2 eps_q (T_k = the k-th largest approximate maximum): the k tiles on top have exact maxima >= T_k - eps, so
// the k-th best score s_k >= T_k - eps, and a tile holding a row >= s_k has an approximate maximum >= T_k - 2 eps.
// Otherwise the flag is raised: the exact first pass runs for this batch.
//
// eps_q, in scaled units (x' = x * 2^-ex, |x'| < 1; q' = q * 2^-eq, |q'| < 1): per component the fp16 rounding is
// |dx'| <= 2^-11 |x'| + 2^-25 (the second term covers fp16's subnormal range), the same for q'.  Hence
//   |x^ . q^ - x' . q'| <= (2^-10 + 2^-22) |x'| |q'| + 2^-25 (|q'|_1 + |x'|_1) (1 + 2^-11) <= ... + d 2^-24 (1 + 2^-11)
// (the products of two fp16 values are exact in the MFMA's fp32, the accumulation of d of them adds d 2^-24 |x'| |q'|;
// so does the accumulation inside the exact kernel this pass is compared with).  |x'| <= R * 2^-ex with R the largest
// row norm.  The comparison only holds if the exact fp32 scores neither overflow nor sink into fp32's subnormal
// range: |ex + eq| <= 100 is required, else the flag is raised.
__global__ __launch_bounds__(64) void dense_hi_check_kernel(const float* __restrict__ vals, int kc1, int k,
                                                            const float* __restrict__ Q, int d, float row_norm_max,
                                                            float x_scale, int x_exp, long n_tiles,
                                                            long long* __restrict__ ids,
                                                            const unsigned int* __restrict__ total, unsigned int cap,
                                                            int* __restrict__ flag, unsigned int* __restrict__ unresolved) {
  const int q = blockIdx.x, lane = threadIdx.x;
  float amax = 0.f;
  bool nan = false;
  for (int j = lane; j < d; j += 64) {
    const float x = Q[(size_t)q * d + j];
    nan |= x != x;
    amax = fmaxf(amax, fabsf(x));
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) amax = fmaxf(amax, __shfl_xor(amax, sft));
  nan = __any(nan);
  int e = 0;
  if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e);
  const float sc = ldexpf(1.f, -e);  // the tile kernel's scale of this query
  float ss = 0.f;
  for (int j = lane; j < d; j += 64) {
    const float x = Q[(size_t)q * d + j] * sc;
    ss += x * x;
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) ss += __shfl_xor(ss, sft);
  const float rel = 1.125f * (9.765625e-4f + 2.4e-7f + 2.f * (float)(d + 8) * 5.9604645e-8f);
  const float eps = rel * sqrtf(ss) * (row_norm_max * x_scale) + 1.125f * (float)d * 5.9604645e-8f;
  const bool bad = nan || !(amax <= FLT_MAX) || !(eps == eps) || e + x_exp > 100 || e + x_exp < -100;
  const float* v = vals + (size_t)q * kc1;
  bool raise = bad;
  // the flat candidate list overflowed (the sample's threshold let too many maxima pass), or the query has fewer than kc1
  // candidates (a NaN threshold: fewer than kc1 sample tiles with a real maximum): the exact chain decides
  if (*total > cap || ids[(size_t)q * kc1 + kc1 - 1] < 0) raise = true;
  float cut = -FLT_MAX;
  if (!raise && n_tiles >= kc1) {  // fewer tiles than candidates: every tile is one already
    const float Tk = v[k - 1], last = v[kc1 - 1];
    cut = Tk - 2.f * eps;
    raise = !(last < cut);
  }
  if (raise) {  // (every lane holds the same values: one lane reports)
    if (lane == 0) {
      if (atomicOr(flag, 1) == 0) atomicAdd(unresolved + 1, 1u);  // passes whose flag went up (the first query to raise it)
      atomicAdd(unresolved, 1u);                                   // queries; both: amdr_dense_hi_counters
    }
    return;
  }
  // Resolved: only the tiles at or above the cut can hold one of the k best rows — the others are dropped from the
  // query's list (the list is sorted by maximum: a suffix), ~14 of 33 stay at k = 10 and the exact pass scores those.
  for (int j = lane; j < kc1; j += 64)
    if (v[j] < cut) ids[(size_t)q * kc1 + j] = -1ll;
}

// d = 1 024: a 48-query tile (96 KiB; 64 queries + the stages would take all 160 KiB of LDS)
bool dense_hi_supported(int d) { return d >= 128 && d <= 1024 && d % 128 == 0; }
int dense_hi_max_queries(int d) { return hi_query_tile(d); }

// Entries of a wave's staging buffer: what the 160 KiB of LDS leave beside the query tile and the stages, at most 384
// (24 KiB for the 8 waves).  A buffer that fills in the middle of a wave's run is flushed entry by entry with returning
// atomics on the per-query counters — 2 048 waves x ~90 entries on 48 addresses at 7.5 M x 1 024 rows, where the buffer
// held 128: the scan took 5.09 ms against 4.71 with the flat list of round 3.  Sized so that a run's entries fit (the
// expected number is kc x sample stride x queries x tiles per wave / tiles = ~90-160), only the block-binned flush at the
// end of the run is left: one atomic per (block, query).
static int hi_wbuf_entries(int d) {
  const char* e = getenv("AMDR_DENSE_HI_WBUF");  // test hook: a 64-entry buffer flushes after every emitting tile
  if (e && atoi(e) >= 64 && atoi(e) <= 128) return atoi(e);
  const long room = 160 * 1024 - 1024 - (long)hi_query_tile(d) * d * 2 - (long)kHiWaves * kHiStageBytes;
  long n = room / ((long)kHiWaves * (long)sizeof(C32)) / 64 * 64;
  n = n > kHiWbufMax ? kHiWbufMax : n;
  return (int)(n < 128 ? 128 : n);
}
static size_t dense_hi_lds(int d, bool emit) {
  return (size_t)hi_query_tile(d) * d * 2 + kHiWaves * kHiStageBytes +
         (emit ? (size_t)kHiWaves * hi_wbuf_entries(d) * sizeof(C32) + 128 * sizeof(int) : 0);
}

template <int D64, bool EMIT>
static int launch_hi(const float* X, long n, const float* Q, int nq, int grid, float* MT, float x_scale, long tile_stride,
                     long n_items, const HiEmit& em, hipStream_t st, int grid_y = 1) {
  const size_t lds = dense_hi_lds(D64 * 64, EMIT);
  AMDR_HIP(hipFuncSetAttribute((const void*)dense_hi_tilemax_kernel<D64, EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  hipLaunchKernelGGL((dense_hi_tilemax_kernel<D64, EMIT>), dim3(grid, grid_y), dim3(kHiWaves * 64), lds, st, X, n, Q, nq, MT,
                     grid_y > 1 ? 64 : hi_mt_cols(nq), x_scale, tile_stride, n_items, em);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

template <bool EMIT>
static int launch_hi_d(const float* X, long n, int d, const float* Q, int nq, float* MT, float x_scale, long tile_stride,
                       const HiEmit& em, hipStream_t st, int grid_y = 1, int scan_tiles = 1) {
  if (!dense_hi_supported(d) || nq < 1 || nq > grid_y * scan_tiles * hi_query_tile(d))
    return fail(AMDR_EINVAL, "dense (fp16 first pass): d=%d nq=%d", d, nq);
  // one persistent block per CU (the query tile fills most of its LDS)
  int dev = 0, cus = 256;
  AMDR_HIP(hipGetDevice(&dev));
  AMDR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const long tiles = (n + 31) / 32;
  const long n_items = (tiles + tile_stride - 1) / tile_stride;
  // the full scan: one persistent block per CU.  The sample (grid_y query tiles in one launch): its items spread over ALL
  // CUs — cus / grid_y blocks per tile, a wave gets one item or none — rather than 8 per block on fewer CUs: it is bound
  // by the bytes its CUs can pull
  long blocks = EMIT ? (n_items + kHiWaves - 1) / kHiWaves : n_items;
  if (blocks > cus / grid_y) blocks = cus / grid_y;
  if (blocks < 1) blocks = 1;
  const int grid = (int)blocks;
  switch (d / 64) {
    case 2: return launch_hi<2, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    case 4: return launch_hi<4, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    case 6: return launch_hi<6, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    case 8: return launch_hi<8, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    case 10: return launch_hi<10, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    case 12: return launch_hi<12, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    case 14: return launch_hi<14, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    case 16: return launch_hi<16, EMIT>(X, n, Q, nq, grid, MT, x_scale, tile_stride, n_items, em, st, grid_y);
    default: return fail(AMDR_EINVAL, "dense (fp16 first pass): unsupported dim %d", d);
  }
}

// the sample: every s-th tile, at most ONE tile per wave of the persistent grid (256 CUs x 8 waves): the pass is bound by
// the latency of a wave's own chunk chain, a second tile on some waves doubled it (2 442 tiles: 56 us)
long dense_hi_sample_stride(long n) {
  const long tiles = (n + 31) / 32;
  const long s = (tiles + 2047) / 2048;
  return s < 1 ? 1 : s;
}
long dense_hi_sample_items(long n) {
  const long tiles = (n + 31) / 32, s = dense_hi_sample_stride(n);
  return (tiles + s - 1) / s;
}
size_t dense_hi_mt_bytes(long n) { return (size_t)dense_hi_sample_items(n) * 64 * sizeof(float); }
// entries of the flat candidate list: 4x the expected m * kc * stride, at least 64 Ki
size_t dense_hi_cand_entries(long n, int m, int kc) {
  size_t e = (size_t)m * kc * dense_hi_sample_stride(n) * 4 + 65536;
  const char* env = getenv("AMDR_DENSE_HI_CAP");  // test hook: a short list overflows -> the exact chain takes over
  if (env && atol(env) >= 64) e = (size_t)atol(env);
  const size_t all = (size_t)((n + 31) / 32) * m;  // never more than every maximum
  return e < all ? e : all + 64;
}

// maxima of the sampled tiles of <= 64 queries: MT[item][cols]
int dense_hi_launch_sample(const float* X, long n, int d, const float* Q, int nq, float* MT, hipStream_t st, float x_scale) {
  HiEmit em{};
  return launch_hi_d<false>(X, n, d, Q, nq, MT, x_scale, dense_hi_sample_stride(n), em, st);
}
// the full scan: every tile maximum >= tau[q * tau_stride] into the flat list (*total zeroed by the caller)
int dense_hi_launch_emit(const float* X, long n, int d, const float* Q, int nq, const float* tau, int tau_stride, void* cand,
                         unsigned int* total, size_t cap, hipStream_t st, float x_scale) {
  if ((n + 31) / 32 >= (1l << kHiQShift)) return fail(AMDR_EINVAL, "dense (fp16 first pass): too many tiles");
  HiEmit em{tau, tau_stride, (C32*)cand, total, (unsigned int)cap, hi_wbuf_entries(d)};
  return launch_hi_d<true>(X, n, d, Q, nq, nullptr, x_scale, 1, em, st);
}
size_t dense_hi_cand_part_bytes(int m, int kc) { return (size_t)kHiCandParts * m * kc * sizeof(C32); }
// step 1 of the candidates' top-kc: part[kHiCandParts][m][kc] (dense_hi_cand_part_bytes); the caller merges the parts
int dense_hi_launch_cand_topk(const void* cand, const unsigned int* total, size_t cap, int m, int kc, void* part,
                              int* nparts, hipStream_t st) {
  const int tcap = topk_cap(kc);
  const size_t lds = (size_t)4 * tcap * sizeof(C32) + 4 * sizeof(int);
  hipLaunchKernelGGL(dense_hi_cand_topk_kernel, dim3(kHiCandParts, m), dim3(256), lds, st, (const C32*)cand, total,
                     (unsigned int)cap, m, kc, tcap, (C32*)part);
  AMDR_HIP(hipGetLastError());
  *nparts = kHiCandParts;
  return AMDR_OK;
}

int dense_hi_launch_transpose(const float* MT, long items, int nq, long ldM, float* M, hipStream_t st) {
  const int tpb = items >= 64 * 1024 ? kHiTrTiles : 64;  // a short sample: more, smaller blocks
  hipLaunchKernelGGL(dense_hi_transpose_kernel, dim3((unsigned)((items + tpb - 1) / tpb)), dim3(256), 0, st, MT, items, nq, ldM,
                     M, tpb);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int dense_hi_launch_check(const float* vals, int64_t* ids, int m, int kc1, int k, const float* Q, int d,
                          float row_norm_max, float x_scale, long n_tiles, const unsigned int* total, size_t cap, int* flag,
                          unsigned int* unresolved, hipStream_t st) {
  int x_exp = 0;
  (void)frexpf(x_scale, &x_exp);  // x_scale = 2^-ex = 0.5 * 2^(1 - ex)
  x_exp = 1 - x_exp;
  hipLaunchKernelGGL(dense_hi_check_kernel, dim3(m), dim3(64), 0, st, vals, kc1, k, Q, d, row_norm_max, x_scale, x_exp, n_tiles,
                     (long long*)ids, total, (unsigned int)cap, flag, unresolved);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

// max |component| and max row norm of X[row0 .. row0 + n): out2 must hold two zeroed unsigned ints
int dense_stats_launch(const float* X, long n, int d, unsigned int* out2, hipStream_t st) {
  if (n <= 0) return AMDR_OK;
  long blocks = (n + 3) / 4;  // a wave per row and round
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dense_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, X, n, d, out2);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}


// =====================================================================================================================
// Round 4: the tail of a large search behind the scan in FOUR launches (+ two gated ones) instead of ~eighteen.
//   sample (all query tiles of the search in one launch) -> dense_hi_tau_kernel -> scan(s), per-query lists
//   -> dense_hi_select_kernel (top-kc of a query's list + the rounding-bound check + its tiles at or above the cut,
//      ascending) -> [gated on the flag: exact tile maxima of the batch, dense_hi_exact_select_kernel for the queries the
//      bound did not resolve] -> dense_rescore_tiles_kernel + final top-k with the column -> row map (dense_mfma.hip).
// At 8 GPUs a rank scans 1.25 M rows in 0.6 ms: the 0.22-0.25 ms of small kernels behind the scan was what capped the
// 1 -> 8 curve.
// =====================================================================================================================

// tau[q] = the kc-th largest sampled tile maximum of query q (a lower bound of the kc-th largest over ALL tiles), NaN
// when the sample holds fewer than kc real maxima.  One wave per query, the <= 2 048 sampled maxima in registers, the
// kc-th largest found bit by bit on the ordered keys (32 x (compare-count + wave sum)); no sort, no transposition pass.
// Also the per-search reset: the query's list counter, and (block 0) the flag and the pass counter.
__global__ __launch_bounds__(64) void dense_hi_tau_kernel(const float* __restrict__ MT, int items, int cols, int qt, int kc,
                                                          float* __restrict__ tau, unsigned int* __restrict__ qcount,
                                                          int* __restrict__ flag, unsigned int* __restrict__ stats,
                                                          int qtiles) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const float* row = MT + (size_t)q * cols;  // query-major: row q, `cols` = its leading dimension (items rounded up to 64)
  (void)qt;
  u32 key[32];
  // unconditional loads on clamped indices: behind `i < items ?` hipcc issued them one at a time, each in its own branch
  // with its own wait — 32 dependent memory round trips, 22 of this kernel's 25 us
  float raw[32];
#pragma unroll
  for (int v = 0; v < 32; ++v) {
    const int i = lane + 64 * v;
    raw[v] = row[i < items ? i : items - 1];
  }
#pragma unroll
  for (int v = 0; v < 32; ++v) key[v] = (lane + 64 * v) < items ? ord32(raw[v]) : 0u;
  // pin the keys in registers: hipcc otherwise re-LOADS them from the (const, restrict) row in every one of the 32
  // rounds below instead of holding 32 registers (the kernel reported 20 VGPRs and took 24 us)
#pragma unroll
  for (int v = 0; v < 32; ++v) asm volatile("" : "+v"(key[v]));
  u32 K = 0u;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    // keys >= cand, counted on the scalar unit: 32 independent compares whose lane masks are popcounted and added as
    // scalars (per-lane counts + a cross-lane sum were a dependent chain of ~110 vector instructions per bit: 13 us)
    const u32 cand = K | (1u << bit);
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < 32; ++v) cnt += __popcll(__ballot(key[v] >= cand));
    if (cnt >= kc) K = cand;
  }
  if (lane == 0) {
    tau[q] = K > 1u ? unord32(K) : __uint_as_float(0x7fc00000u);  // key 1 = a NaN maximum, 0 = nothing
    qcount[q] = 0u;
    if (q == 0) {
      *flag = 0;
      // passes through the fp16 first pass, in query tiles (the unit the width adaptation was tuned on), counted where
      // the flags are counted (a hipGraph replay bumps both)
      atomicAdd(stats + 4, (unsigned int)qtiles);
    }
  }
}

template <int V>
__device__ __forceinline__ int select_list(const C32* __restrict__ src, unsigned int n, int k, int lane, C32* buf) {
  C32 keys[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const unsigned int i = lane + 64u * v;
    keys[v] = i < n ? src[i] : C32::pad();
  }
  return wave_select_small<C32, V>(keys, k, buf + 64, lane, buf);
}

// One block per query: (1) the kc best entries of the query's candidate list (score desc, tile asc — the order of every
// top-k here); (2) the check of dense_hi_check_kernel: is every tile that can hold one of the k best rows among them?
// (3) resolved -> the tiles at or above the cut T_k - 2 eps, ASCENDING, into list[q * kc ..], count[q]; unresolved ->
// count[q] = 0, unres[q] = 1, the flag raised: the gated exact pass fills the query's list instead.
__global__ __launch_bounds__(256) void dense_hi_select_kernel(const C32* __restrict__ qlist,
                                                              const unsigned int* __restrict__ qcount, unsigned int qcap,
                                                              int kc1, int k, int tcap, const float* __restrict__ Q, int d,
                                                              float row_norm_max, float x_scale, int x_exp, long n_tiles,
                                                              int* __restrict__ list, int* __restrict__ count,
                                                              int* __restrict__ unres, int* __restrict__ flag,
                                                              unsigned int* __restrict__ unresolved, int qtiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)4 * tcap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x;
  const unsigned int have = qcount[q];
  const unsigned int n = have < qcap ? have : qcap;
  const C32* src = qlist + (size_t)q * qcap;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * tcap, tcap, kc1);
  // Short lists (the usual case: ~kc x sample stride entries) are ranked by ONE wave in registers (topk.hpp
  // wave_select_small: lane bests, their kc-th as the cut, the few survivors sorted) — the staged selector's LDS sorts
  // (two or three per wave, then three merges) were 25 of this kernel's 32 us.  Mass ties at the cut (> 64 survivors),
  // long lists and wide cuts take the staged selector on all four waves.
  int got = -1;
  if (n <= 4096u && kc1 <= 64) {  // block-uniform
    if (wave != 0) return;
    if (n <= 512u)
      got = select_list<8>(src, n, kc1, lane, tk.buf);
    else if (n <= 1024u)
      got = select_list<16>(src, n, kc1, lane, tk.buf);
    else if (n <= 2048u)
      got = select_list<32>(src, n, kc1, lane, tk.buf);
    else  // (a four-tile pass samples every 77th tile of a 1.25 M-row shard: ~2 500 entries per query)
      got = select_list<64>(src, n, kc1, lane, tk.buf);
    if (got >= 0) {
      tk.cnt = got;
    } else {  // mass ties: this wave alone, staged
      for (unsigned int base = 0; base < n; base += 64) {
        const unsigned int i = base + lane;
        const C32 e = i < n ? src[i] : C32::pad();
        tk.push_lanes(e, !e.is_pad(), lane);
      }
      tk.finalize(lane);
    }
  } else {
    constexpr int UN = 4;  // entries per lane requested before the first is looked at
    for (unsigned int base = (unsigned int)wave * 64; base < n; base += 256 * UN) {
      C32 e[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const unsigned int i = base + 256 * u + lane;
        e[u] = i < n ? src[i] : C32::pad();
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) tk.push_lanes(e[u], !e[u].is_pad(), lane);
    }
    tk.finalize(lane);
    block_combine_topk(tk, lists, tcap, 4, wave, lane, cnts);
    if (wave != 0) return;
  }
  // ---- the rounding bound of this query (dense_hi_check_kernel states it)
  float amax = 0.f;
  bool nan = false;
  for (int j = lane; j < d; j += 64) {
    const float x = Q[(size_t)q * d + j];
    nan |= x != x;
    amax = fmaxf(amax, fabsf(x));
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) amax = fmaxf(amax, __uint_as_float(lane_xor_sw(__float_as_uint(amax), sft)));
  nan = __any(nan);
  int e = 0;
  if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e);
  const float sc = ldexpf(1.f, -e);
  float ss = 0.f;
  for (int j = lane; j < d; j += 64) {
    const float x = Q[(size_t)q * d + j] * sc;
    ss += x * x;
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) ss += __uint_as_float(lane_xor_sw(__float_as_uint(ss), sft));
  const float rel = 1.125f * (9.765625e-4f + 2.4e-7f + 2.f * (float)(d + 8) * 5.9604645e-8f);
  const float eps = rel * sqrtf(ss) * (row_norm_max * x_scale) + 1.125f * (float)d * 5.9604645e-8f;
  const bool bad = nan || !(amax <= FLT_MAX) || !(eps == eps) || e + x_exp > 100 || e + x_exp < -100;
  const int cnt = tk.cnt;
  const long want = n_tiles < kc1 ? n_tiles : (long)kc1;
  bool raise = bad || have > qcap || cnt < want;  // an overflowed list or too few candidates (NaN threshold): the exact chain decides
  float cut = -FLT_MAX;
  if (!raise && n_tiles >= kc1) {  // fewer tiles than candidates: every tile is one already
    const float Tk = tk.buf[k - 1].score(), last = tk.buf[kc1 - 1].score();
    cut = Tk - 2.f * eps;
    raise = !(last < cut);
  }
  if (raise) {
    if (lane == 0) {
      count[q] = 0;
      unres[q] = 1;
      if (atomicOr(flag, 1) == 0) atomicAdd(unresolved + 1, (unsigned int)qtiles);  // passes (query tiles) whose flag went up
      atomicAdd(unresolved, 1u);                                                     // queries
    }
    return;
  }
  // the tiles at or above the cut (the list is sorted by maximum: a prefix), ascending by tile: rank = smaller tiles
  int keep = 0;
  for (int j0 = 0; j0 < cnt; j0 += 64) {
    const int j = j0 + lane;
    keep += __popcll(__ballot(j < cnt && !(tk.buf[j].score() < cut)));
  }
  for (int j = lane; j < keep; j += 64) {
    const long long t = tk.buf[j].id();
    int rank = 0;
    for (int i = 0; i < keep; ++i) rank += tk.buf[i].id() < t ? 1 : 0;
    list[(size_t)q * kc1 + rank] = (int)t;
  }
  if (lane == 0) {
    count[q] = keep;
    unres[q] = 0;
  }
}

// Gated (the flag of the pass) and per query (unres[q]): the k tiles with the largest EXACT maxima M[q][tile] (fp32 matrix
// instructions, dense_mfma.hip mode 1) — step 2 of the exact two-level form — ascending into the query's list.  The exact
// re-scoring and the final top-k then treat every query of the batch alike.
__global__ __launch_bounds__(256) void dense_hi_exact_select_kernel(const float* __restrict__ M, long ldM, long n_tiles, int k,
                                                                    int tcap, int list_stride, int* __restrict__ list,
                                                                    int* __restrict__ count, const int* __restrict__ unres,
                                                                    const int* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (*gate == 0) return;
  const int q = blockIdx.x;
  if (unres[q] == 0) return;
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)4 * tcap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* row = M + (size_t)q * ldM;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * tcap, tcap, k);
  for (long base = (long)wave * 256; base < n_tiles; base += 4 * 256) {  // ldM is a multiple of 32 floats: whole float4s
    const long r0 = base + 4 * lane;
    const hi4f z = {0.f, 0.f, 0.f, 0.f};
    const hi4f x = r0 < n_tiles ? *reinterpret_cast<const hi4f*>(row + r0) : z;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long r = r0 + e;
      const bool v = r < n_tiles;
      tk.push_lanes(v ? C32::make(x[e], (u32)r) : C32::pad(), v, lane);
    }
  }
  tk.finalize(lane);
  block_combine_topk(tk, lists, tcap, 4, wave, lane, cnts);
  if (wave != 0) return;
  const int keep = tk.cnt;
  for (int j = lane; j < keep; j += 64) {
    const long long t = tk.buf[j].id();
    int rank = 0;
    for (int i = 0; i < keep; ++i) rank += tk.buf[i].id() < t ? 1 : 0;
    list[(size_t)q * list_stride + rank] = (int)t;
  }
  if (lane == 0) count[q] = keep;
}

// ---- host side of the round-4 tail -------------------------------------------------------------------------------------
// the sample of a search with `qtiles` query tiles: 2 048 / qtiles items per tile (one item per wave of the grid, all
// tiles in ONE launch)
long dense_hi2_sample_stride(long n, int qtiles) {
  const long tiles = (n + 31) / 32;
  // The sample READS its tiles (96 KiB each at d = 768: 2 048 tiles = 197 MB, 30-47 us — bandwidth, not latency), and a
  // coarser sample only loosens the threshold: kc x stride entries per query reach the lists.  About every 32nd tile, at
  // least 512 and at most 2 048 tiles in all: a 1.25 M-row shard samples 1 221 tiles (~1 050 entries per query at k = 10:
  // the register selector's range), 10 M rows 2 048.
  long total = tiles / 32;
  total = total < 512 ? 512 : (total > 2048 ? 2048 : total);
  long per = total / (qtiles < 1 ? 1 : qtiles);
  per = per / kHiWaves * kHiWaves;
  if (per < 512) per = 512;  // per query tile: well above the widest cut (kc <= 256), or the kc-th sampled maximum does not exist
  const long s = (tiles + per - 1) / per;
  return s < 1 ? 1 : s;
}
long dense_hi2_sample_items(long n, int qtiles) {
  const long tiles = (n + 31) / 32, s = dense_hi2_sample_stride(n, qtiles);
  return (tiles + s - 1) / s;
}
// entries of ONE query's candidate list: 4x the expected kc * stride, at least 1 024; never more than every tile
size_t dense_hi2_qcap(long n, int qtiles, int kc) {
  size_t e = (size_t)kc * dense_hi2_sample_stride(n, qtiles) * 4 + 1024;
  const char* env = getenv("AMDR_DENSE_HI_CAP");  // test hook (entries per 64 queries): short lists overflow -> the exact chain takes over
  if (env && atol(env) >= 64) e = (size_t)atol(env) / 64;
  const size_t all = (size_t)((n + 31) / 32);
  return e < all ? e : all;
}
long dense_hi2_sample_ld(long n, int qtiles) { return (dense_hi2_sample_items(n, qtiles) + 63) / 64 * 64; }
int dense_hi2_launch_sample(const float* X, long n, int d, const float* Q, int nq, int qtiles, float* MT, hipStream_t st,
                            float x_scale) {
  HiEmit em{};
  em.tau_stride = -(int)dense_hi2_sample_ld(n, qtiles);  // query-major maxima: MT[query][ld]
  return launch_hi_d<false>(X, n, d, Q, nq, MT, x_scale, dense_hi2_sample_stride(n, qtiles), em, st, qtiles);
}
int dense_hi2_launch_tau(const float* MT, long n, int d, int nq, int qtiles, int kc, float* tau, unsigned int* qcount,
                         int* flag, unsigned int* stats, hipStream_t st) {
  const int items = (int)dense_hi2_sample_items(n, qtiles);
  if (items > 2048) return fail(AMDR_EINVAL, "dense (fp16 first pass): %d sample items", items);
  hipLaunchKernelGGL(dense_hi_tau_kernel, dim3(nq), dim3(64), 0, st, MT, items, (int)dense_hi2_sample_ld(n, qtiles),
                     hi_query_tile(d), kc, tau, qcount, flag, stats, qtiles);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}
// the scan of ONE query tile: maxima >= tau[q] into the per-query lists
int dense_hi2_launch_emit(const float* X, long n, int d, const float* Q, int nq, const float* tau, void* qlist,
                          unsigned int* qcount, size_t qcap, hipStream_t st, float x_scale, int qtiles) {
  if ((n + 31) / 32 >= (1l << kHiQShift)) return fail(AMDR_EINVAL, "dense (fp16 first pass): too many tiles");
  HiEmit em{tau, 1, nullptr, nullptr, 0u, hi_wbuf_entries(d), (C32*)qlist, qcount, (unsigned int)qcap, qtiles};
  return launch_hi_d<true>(X, n, d, Q, nq, nullptr, x_scale, 1, em, st, 1, qtiles);
}
int dense_hi2_launch_select(const void* qlist, const unsigned int* qcount, size_t qcap, int m, int kc, int k, const float* Q,
                            int d, float row_norm_max, float x_scale, long n_tiles, int* list, int* count, int* unres,
                            int* flag, unsigned int* unresolved, hipStream_t st) {
  int x_exp = 0;
  (void)frexpf(x_scale, &x_exp);  // x_scale = 2^-ex = 0.5 * 2^(1 - ex)
  x_exp = 1 - x_exp;
  const int tcap = topk_cap(kc);
  const size_t lds = (size_t)4 * tcap * sizeof(C32) + 4 * sizeof(int);
  hipLaunchKernelGGL(dense_hi_select_kernel, dim3(m), dim3(256), lds, st, (const C32*)qlist, qcount, (unsigned int)qcap, kc, k,
                     tcap, Q, d, row_norm_max, x_scale, x_exp, n_tiles, list, count, unres, flag, unresolved,
                     (m + hi_query_tile(d) - 1) / hi_query_tile(d));
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}
int dense_hi2_launch_exact_select(const float* M, long ldM, long n_tiles, int m, int k, int list_stride, int* list, int* count,
                                  const int* unres, const int* gate, hipStream_t st) {
  const int tcap = topk_cap(k);
  const size_t lds = (size_t)4 * tcap * sizeof(C32) + 4 * sizeof(int);
  hipLaunchKernelGGL(dense_hi_exact_select_kernel, dim3(m), dim3(256), lds, st, M, ldM, n_tiles, k, tcap, list_stride, list,
                     count, unres, gate);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

}  // namespace amdr
