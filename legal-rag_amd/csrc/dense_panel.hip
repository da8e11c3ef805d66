// Dense channel, long-batch form: S[query][row] = <Q[query], X[row]> for hundreds to
// hundreds of thousands of queries per call (the throughput path of
// HybridRetriever.search_batch; same contract as dense.hip — exact inner products behind
// faiss `index.search`, legalrag/retrieval/dense_retriever.py:42).
//
// With that many queries the product is bound by the fp32 matrix pipe, and what decides how
// close a kernel gets to it is how few bytes each MFMA pulls through the vector-memory path
// and the LDS.  The 32 x 32 wave tile of dense_mfma.hip streams 4 KiB through L2 -> registers
// -> LDS -> registers per 32 MFMAs (measured: its non-MFMA stream alone takes longer than its
// MFMAs).  Here a block of WAVES waves shares one PANEL of NB x 16 chunk rows, streamed
// through LDS in 32-float K chunks, and each wave owns 32 queries:
//     wave tile   32 queries x 16*NB rows, 2*NB accumulator blocks of 16 x 16 (v_mfma_f32_16x16x4_f32)
//     per chunk   8 k-steps x 2*NB MFMAs per wave; LDS reads 2*NB + 4 ds_read_b128; global -> LDS
//                 by LDS-DMA (global_load_lds_dwordx4): 4 KiB of the wave's own queries + its
//                 share (2*NB/WAVES KiB) of the panel chunk — no staging registers, no ds_write
// so NB = 6 moves 7 KiB per 96 MFMAs instead of 4 KiB per 32, and reads 16 fragments from LDS
// per 96 MFMAs instead of 8 (+4 writes) per 32.
//   A operand (16 x 4) = panel rows   : lane (i16 = l & 15, kq = l >> 4) <- row 16 b + i16
//   B operand (4 x 16) = query rows   : lane (i16, kq)                  <- query 16 bq + i16
// so the 16 x 16 result block has the QUERY on the lane and four consecutive chunk rows in the
// lane's four registers: S[query][row .. row+3] is one 16-byte store per block, no transpose.
// Inside a 32-float chunk lane group kq takes the 16-byte slots kq and 4 + kq of both
// operands' rows — a permutation of k the dot product cannot see.
// LDS images are lane-linear per DMA piece (8 rows x 128 B); the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and to
// the read address (the destination of an LDS-DMA is base + lane * 16, never swizzled).
// Pipeline: two LDS buffers; the DMA of chunk c+1 is issued before the fragment reads of chunk
// c and retired (vmcnt(0)) at the block barrier that ends chunk c.
#include "common.hpp"

#include <atomic>
#include <cstdlib>

namespace amdr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AMDR_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AMDR_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int kPanelKC = 32;  // floats of K per chunk: 128 B = 8 slots of 16 B per row

// byte offset of logical 16-B slot `slot` of row `row` in a [rows][128 B] image
__device__ __forceinline__ int panel_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// One (32 queries per wave) x (NBP x 16 rows) tile.  NBE = row blocks the panel buffers are
// sized and filled for (NBP rounded up so that every wave issues the same number of DMA
// pieces: the surplus rows repeat the last valid row and are never multiplied).
// Tried and dropped (same-box A/B, scripts/ab_dense_panel.py): (i) ONE LDS buffer for the wave's
// own query chunk (fragments to registers first, then the next DMA over them: 40 KiB per block,
// four blocks per CU instead of two) — 3-8 % slower at every shape; (ii) fragment reads pinned one
// half-chunk ahead of their MFMAs across the barrier, DMA two chunks ahead spread between the MFMA
// groups (sched_barrier-pinned phases) — 2-9 % slower than hipcc's own schedule of this loop.
template <int NBP, int NBE, int WAVES>
__device__ __forceinline__ void panel_tile(const float* __restrict__ X, long n, const float* __restrict__ Q, int nq,
                                           int d, long row0, int q0w, long ldS, float* __restrict__ S,
                                           unsigned char* smem, int lane, int wave) {
  constexpr int XBYTES = NBE * 16 * 128;          // one panel chunk
  constexpr int PPW = 2 * NBE / WAVES;            // panel DMA pieces per wave per chunk
  static_assert((2 * NBE) % WAVES == 0, "panel pieces must divide over the waves");
  unsigned char* xs = smem;                                  // [2][XBYTES]
  unsigned char* qs = smem + 2 * XBYTES + wave * 2 * 4096;   // [2][4096], private to the wave
  const int i16 = lane & 15, kq = lane >> 4;
  const int prow = lane >> 3, pslot = lane & 7;  // DMA role: row in the 8-row piece, PHYSICAL slot
  const int nch = d / kPanelKC;

  // per-lane source pointers of the DMA pieces (chunk 0); a piece is 8 rows x 128 B
  const float* xp[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int r = 8 * (wave + WAVES * j) + prow;  // row in the panel
    long gr = row0 + r;
    if (gr > n - 1) gr = n - 1;
    xp[j] = X + (size_t)gr * d + ((pslot ^ ((r >> 1) & 7)) << 2);
  }
  const float* qp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 8 * j + prow;
    int gq = q0w + r;
    if (gq > nq - 1) gq = nq - 1;
    qp[j] = Q + (size_t)gq * d + ((pslot ^ ((r >> 1) & 7)) << 2);
  }
  // lane-dependent part of the fragment read addresses (slot 4u + kq of row i16 of a block)
  int foff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) foff[u] = panel_off(i16, 4 * u + kq);

  f32x4 acc[NBP][2];
#pragma unroll
  for (int b = 0; b < NBP; ++b)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) acc[b][bq] = f32x4{0.f, 0.f, 0.f, 0.f};

// DMA piece J of chunk C (J < PPW: this wave's share of the panel; then its 4 query pieces)
#define AMDR_PANEL_PIECE(C, BUF, J)                                                                      \
  {                                                                                                      \
    if ((J) < PPW)                                                                                       \
      __builtin_amdgcn_global_load_lds(AMDR_GPTR(xp[(J) < PPW ? (J) : 0] + (size_t)(C) * kPanelKC),      \
                                       AMDR_LPTR(xs + (BUF) * XBYTES + (wave + WAVES * (J)) * 1024), 16, 0, 0); \
    else                                                                                                 \
      __builtin_amdgcn_global_load_lds(AMDR_GPTR(qp[(J) >= PPW ? (J) - PPW : 0] + (size_t)(C) * kPanelKC), \
                                       AMDR_LPTR(qs + (BUF) * 4096 + ((J) - PPW) * 1024), 16, 0, 0);     \
  }
#define AMDR_PANEL_ISSUE(C, BUF, J0, J1) \
  { _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) AMDR_PANEL_PIECE(C, BUF, j_) }
#define AMDR_PANEL_READ(XF, QF, BUF, U)                                                                  \
  {                                                                                                      \
    _Pragma("unroll") for (int bq = 0; bq < 2; ++bq) QF[bq] =                                            \
        *reinterpret_cast<const f32x4*>(qs + (BUF) * 4096 + bq * 2048 + foff[U]);                        \
    _Pragma("unroll") for (int b = 0; b < NBP; ++b) XF[b] =                                              \
        *reinterpret_cast<const f32x4*>(xs + (BUF) * XBYTES + b * 2048 + foff[U]);                       \
  }
#define AMDR_PANEL_MFMA(XF, QF, E0, E1)                                                                  \
  _Pragma("unroll") for (int e = (E0); e < (E1); ++e) _Pragma("unroll") for (int b = 0; b < NBP; ++b)    \
      _Pragma("unroll") for (int bq = 0; bq < 2; ++bq) acc[b][bq] =                                      \
          __builtin_amdgcn_mfma_f32_16x16x4f32(XF[b][e], QF[bq][e], acc[b][bq], 0, 0, 0);

  constexpr int NP = PPW + 4;  // DMA pieces per wave per chunk
  // An LDS-DMA is ordered for a later ds_read only by the ISSUING wave's vmcnt wait followed by a
  // barrier the reader has passed.  hipcc puts that wait in front of a __syncthreads() only when it
  // sees the DMA in straight-line code before it (it did not once the DMA sat behind the loop's
  // back edge), so the wait is written out.
#define AMDR_PANEL_PUBLISH()                          \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
  __syncthreads();
  {
    AMDR_PANEL_ISSUE(0, 0, 0, NP)
    AMDR_PANEL_PUBLISH()  // chunk 0 is in LDS for every wave
    for (int c = 0; c < nch; ++c) {
      const int buf = c & 1;
      if (c + 1 < nch) AMDR_PANEL_ISSUE(c + 1, buf ^ 1, 0, NP)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 xf[NBP], qf[2];
        AMDR_PANEL_READ(xf, qf, buf, u)
        AMDR_PANEL_MFMA(xf, qf, 0, 4)
      }
      AMDR_PANEL_PUBLISH()  // every wave has read chunk c; chunk c+1 has landed
    }
  }
#undef AMDR_PANEL_PIECE
#undef AMDR_PANEL_PUBLISH
#undef AMDR_PANEL_ISSUE
#undef AMDR_PANEL_READ
#undef AMDR_PANEL_MFMA
  // acc[b][bq][r] = <X[row0 + 16 b + 4 kq + r], Q[q0w + 16 bq + i16]>
#pragma unroll
  for (int bq = 0; bq < 2; ++bq) {
    const int q = q0w + 16 * bq + i16;
    if (q < nq) {
      float* srow = S + (size_t)q * ldS + row0 + 4 * kq;
#pragma unroll
      for (int b = 0; b < NBP; ++b) *reinterpret_cast<f32x4*>(srow + 16 * b) = acc[b][bq];
    }
  }
}

constexpr int panel_nbe(int nb, int waves) { return (2 * nb + waves - 1) / waves * waves / 2; }

// grid: 1-D, m_tiles x parts LOGICAL blocks.  Part p covers `base` (+1 for p < rem) row blocks of 16.
// Logical order (after the XCD remap): groups of `gm` query tiles; inside a group part-major,
// so that the blocks sharing a panel are neighbours on one XCD and a group's query tiles stay
// in that XCD's L2 while its parts go by.
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dense_panel_scores_kernel(const float* __restrict__ X, long n,
                                                                        const float* __restrict__ Q, int nq, int d,
                                                                        int parts, int base, int rem, int m_tiles,
                                                                        int gm, long ldS, float* __restrict__ S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // A block walks the logical blocks bid = blockIdx.x, + gridDim.x, ... (the host launches either
  // one block per logical block or, persistent form, as many as stay resident; gridDim.x % 8 == 0
  // there, so bid & 7 is still the XCD the block runs on).  The last barrier of a tile is behind
  // every wave's last fragment read, so the next tile's first DMA may overwrite the buffers.
  // (Tried: the tiles of a block as ONE chunk stream — the next tile's first chunk issued in place
  // of "chunk nch", so that it lands during the last MFMAs and the epilogue: +- 0 at every shape.)
  const int nwg = m_tiles * parts;
  for (int bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
    int part, mt;
    {
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
      const int per_group = gm * parts;
      const int g = logical / per_group, rr = logical - g * per_group;
      int gsz = m_tiles - g * gm;
      if (gsz > gm) gsz = gm;
      part = rr / gsz;
      mt = g * gm + (rr - part * gsz);
    }
    const int nbp = base + (part < rem ? 1 : 0);
    const long row0 = 16L * ((long)part * base + (part < rem ? part : rem));
    const int q0w = (mt * WAVES + wave) * 32;
    constexpr int NBE = panel_nbe(NB, WAVES);
    if (nbp == NB)
      panel_tile<NB, NBE, WAVES>(X, n, Q, nq, d, row0, q0w, ldS, S, smem, lane, wave);
    else if (NB > 1)
      panel_tile<(NB > 1 ? NB - 1 : 1), NBE, WAVES>(X, n, Q, nq, d, row0, q0w, ldS, S, smem, lane, wave);
  }
}

constexpr int kPanelWaves = 4;
constexpr int kPanelNBMax = 8;

static size_t panel_lds(int nb, int waves) { return (size_t)2 * panel_nbe(nb, waves) * 2048 + (size_t)waves * 8192; }

bool dense_panel_supported(long n, int d, int nq) {
  const char* pin = getenv("AMDR_DENSE_PANEL");
  if (pin && pin[0] == '0') return false;
  if (d < kPanelKC || d % kPanelKC != 0 || n < 1) return false;
  if (pin && pin[0] == '1') return nq >= 1;
  return nq >= 3 * 32;  // at least three of a block's four waves have queries
}

void dense_panel_plan(long n, int d, int nq, DensePanelPlan* p) {
  const long nb_total = (n + 15) / 16;
  const int m_tiles = ceil_div(nq, 32 * kPanelWaves);
  // parts: every part has NB or NB - 1 row blocks, NB <= kPanelNBMax; blocks = m_tiles x parts are
  // dealt over 256 CUs, two resident per CU (LDS).  Fitted on measured launches (591 / 1260 / 1851
  // rows x 9 344 ... 37 376 queries, profiles/r02_dense_panel_ab.md): a block costs ~(row blocks +
  // 0.2); a CU works its blocks in pairs, and a last block left alone on its CU runs at 0.6 of the
  // paired throughput — so the cut is chosen to give every CU an even number of blocks.
  long pmin = (nb_total + kPanelNBMax - 1) / kPanelNBMax;
  long pmax = nb_total < 4 * pmin + 8 ? nb_total : 4 * pmin + 8;
  if (pmax < pmin) pmax = pmin;
  long best_p = pmin;
  double best = 1e300;
  const char* pin = getenv("AMDR_PANEL_PARTS");
  if (pin && atol(pin) >= pmin && atol(pin) <= nb_total) {
    best_p = atol(pin);
  } else {
    for (long q = pmin; q <= pmax; ++q) {
      const double bpc = (double)m_tiles * (double)q / 256.0;
      const long top = (long)(bpc + 0.999999);
      double units = (double)top;
      if (top & 1) units += 0.6 * (top > bpc ? bpc - (double)(top - 1) : 1.0);
      const double est = units * ((double)nb_total / (double)q + 0.2);
      if (est < best * (1.0 - 1e-9)) {
        best = est;
        best_p = q;
      }
    }
  }
  p->parts = (int)best_p;
  p->base = (int)(nb_total / best_p);
  p->rem = (int)(nb_total % best_p);
  p->nb = p->base + (p->rem ? 1 : 0);
  p->m_tiles = m_tiles;
  p->gm = 8;
  if (const char* g = getenv("AMDR_PANEL_GM")) {
    const int v = atoi(g);
    if (v >= 1 && v <= 4096) p->gm = v;
  }
  p->waves = kPanelWaves;
  p->lds = panel_lds(p->nb, kPanelWaves);
}

template <int NB>
static int launch_panel(const DensePanelPlan& p, const float* X, long n, int d, const float* Q, int nq, long ldS,
                          float* S, hipStream_t st) {
  // > 64 KiB of dynamic LDS needs the opt-in; it is a per-device function attribute, cheap to set
  AMDR_HIP(hipFuncSetAttribute((const void*)dense_panel_scores_kernel<NB, kPanelWaves>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds(NB, kPanelWaves)));
  // Persistent grid: as many blocks as stay resident at once (occupancy x CUs, a multiple of 8 so that a
  // block's logical ids keep its XCD), each walking its share of the logical blocks — no block is launched
  // into the LDS a retiring one frees, and the tile loop runs on without a dispatch in between (UCC-en step:
  // 280 -> 268 us; other shapes + 0.5-1.2 %).  AMDR_PANEL_PERSIST=0 launches one block per logical block.
  int grid = p.m_tiles * p.parts;
  const char* pe = getenv("AMDR_PANEL_PERSIST");
  if (!(pe && pe[0] == '0')) {
    // (device, NB) -> resident blocks per CU x CUs, asked once per process and device (two threads that race here
    // store the same value; the slot is an atomic so that neither reads a torn one)
    static std::atomic<int> resident_cache[64];
    int dev = 0, cus = 0, per_cu = 0;
    AMDR_HIP(hipGetDevice(&dev));
    const int slot = dev & 63;
    int cached = resident_cache[slot].load(std::memory_order_relaxed);
    if (cached == 0) {
      AMDR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      AMDR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dense_panel_scores_kernel<NB, kPanelWaves>,
                                                            kPanelWaves * 64, p.lds));
      cached = (per_cu << 16) | (cus & 0xffff);
      resident_cache[slot].store(cached, std::memory_order_relaxed);
    }
    per_cu = cached >> 16;
    cus = cached & 0xffff;
    if (pe && atoi(pe) >= 1) per_cu = atoi(pe);
    const int resident = per_cu * cus / 8 * 8;
    if (resident >= 8 && grid > resident) grid = resident;
  }
  hipLaunchKernelGGL((dense_panel_scores_kernel<NB, kPanelWaves>), dim3(grid),
                     dim3(kPanelWaves * 64), p.lds, st, X, n, Q, nq, d, p.parts, p.base, p.rem, p.m_tiles, p.gm, ldS, S);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}
int dense_panel_launch_scores(const DensePanelPlan& p, const float* X, long n, int d, const float* Q, int nq,
                              long ldS, float* S, hipStream_t st) {
  switch (p.nb) {
    case 1: return launch_panel<1>(p, X, n, d, Q, nq, ldS, S, st);
    case 2: return launch_panel<2>(p, X, n, d, Q, nq, ldS, S, st);
    case 3: return launch_panel<3>(p, X, n, d, Q, nq, ldS, S, st);
    case 4: return launch_panel<4>(p, X, n, d, Q, nq, ldS, S, st);
    case 5: return launch_panel<5>(p, X, n, d, Q, nq, ldS, S, st);
    case 6: return launch_panel<6>(p, X, n, d, Q, nq, ldS, S, st);
    case 7: return launch_panel<7>(p, X, n, d, Q, nq, ldS, S, st);
    case 8: return launch_panel<8>(p, X, n, d, Q, nq, ldS, S, st);
    default: return fail(AMDR_EINVAL, "dense (panel): bad plan nb=%d", p.nb);
  }
}

}  // namespace amdr
