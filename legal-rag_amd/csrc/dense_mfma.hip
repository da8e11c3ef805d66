// Dense channel, batched form: 32 queries share one pass over the chunk matrix (5-95 queries per
// call, and the 32-queries-per-pass scan of a large matrix; batches of >= 96 queries take
// dense_panel.hip).
//
// Same contract as dense.hip (exact inner-product top-k behind faiss
// `index.search`, legalrag/retrieval/dense_retriever.py:42) for query batches.
// With B queries per pass the scan is a [n x d] . [d x B] product; beyond ~8
// queries the per-(row, query) cross-lane reductions of the GEMV form saturate
// the vector ALU before HBM, so the batch is tiled 32 queries x 32 rows per wave on the
// fp32-input matrix instruction v_mfma_f32_16x16x4_f32 (four 16x16 accumulator blocks): exact
// fp32 products and accumulation (no TF32/bf16 shortcut exists or is wanted), the same peak
// rate as the vector ALU, but the accumulate needs no cross-lane traffic.  (The tile was first
// built on v_mfma_f32_32x32x2_f32 — same flops per cycle on paper; the 16x16 form measured
// 7.5 % faster on the UCC-en launch in a same-box A/B: the chip holds a higher clock on it.)
//   A (16 x 4): lane (i16 = l&15, kq = l>>4) <- streamed row 16b + i16    (HBM -> registers -> LDS -> registers)
//   B (4 x 16): lane (i16, kq)              <- LDS tile (query) row 16b + i16   (staged once per block)
// The streamed operand is fetched with fully coalesced 16-B/lane loads (8 rows x 128 B per wave
// instruction, each byte read from HBM exactly once), parked in a wave-private, XOR-swizzled
// 4-KiB LDS stage and read back as fragments; four 32-float chunks per wave (128 KiB per CU
// with 8 waves) are always in flight in registers.  The k order inside a chunk is permuted
// (lane group kq takes 16-B slots kq and 4 + kq); both operands use the same permutation,
// which a dot product cannot see.  The finished 32x32 tile leaves as 16-byte stores
// S[query][4 consecutive rows] straight from the accumulators (epilogue).  Top-k is a
// second, slab-parallel pass over S (+8 % traffic at d = 768: 128 B written and read per
// 3072-B row per 32 queries).
#include "common.hpp"
#include "topk.hpp"

#include <cfloat>
#include <cstdlib>

namespace amdr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));  // native vector: stays in registers (HIP's float4 struct did not)

constexpr int kBW = 4;          // waves per block
constexpr int kKC = 32;         // floats of every row per staged chunk (128 B = 8 slots of 16 B)
constexpr int kStageBytes = 32 * kKC * 4;  // 4 KiB: one 32-row x 32-float chunk
#ifndef AMDR_STAGE_BUFS
#define AMDR_STAGE_BUFS 1
#endif
// LDS stages per wave.  One is enough: the LDS unit serves a wave's operations in order, so the
// ds_writes of chunk c+2 cannot overtake the fragment reads of chunk c+1 issued before them, and
// those reads land in registers (FX/FQ) a whole chunk of MFMAs before they are used.
constexpr int kStageBufs = AMDR_STAGE_BUFS;
constexpr int kPieces = kStageBytes / 1024;  // 1-KiB wave loads per chunk (4): 8 rows x 128 B each
constexpr int kDepth = 4;                      // chunks in flight from HBM per wave (16 KiB)

// LDS image of a staged chunk: row r (0..31) at byte r*128 — two rows share one 256-B bank
// row — with its logical 16-B slot s (0..7) at physical slot s ^ ((r >> 1) & 7).  A
// ds_read_b128 lane group covers 16 different rows at the same logical slot: 8 distinct
// physical slots x the 2 halves of the bank row = conflict-free; a ds_write_b128 lane group
// (8 contiguous lanes) writes the 8 slots of one row.
__device__ __forceinline__ int stage_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// The three pipeline steps are macros, not functions: passing the register arrays by
// reference made hipcc (ROCm 7.2) keep them in scratch memory.
#define AMDR_STAGE_CHUNK(ST, G)                                                                   \
  _Pragma("unroll") for (int p_ = 0; p_ < kPieces; ++p_)                                          \
      *reinterpret_cast<v4f*>((ST) + stage_off(8 * p_ + lrow, lslot)) = G[p_];
// Fragments of one 32-float chunk.  v_mfma_f32_16x16x4_f32 takes A[row l&15][k = l>>4] and
// B[k = l>>4][col l&15], one float per lane, so a lane (i16 = l & 15, kq = l >> 4) owns, for each
// of the two 16-row blocks b of the tile, the 16-B slots 4u + kq (u = 0, 1) of row 16b + i16: the
// chunk's eight k-steps take component (u, x..w) of those slots from the four kq groups — a
// permutation of k that both operands share.  Slot choice 4u + kq keeps both images
// conflict-free: a ds_read_b128 lane group {0-3, 12-15, 20-27} (and its three siblings) holds 16
// distinct rows, eight at slot 4u + kq and eight at 4u + (kq ^ 1), and the row swizzles map
// exactly those two sets onto disjoint bank quads.
#define AMDR_READ_FRAGS(ST, C, FX, FQ)                                                            \
  _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_) {                                              \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                            \
      FX[b_][u_] = *reinterpret_cast<const v4f*>((ST) + b_ * 2048 + xoff[u_]);                    \
      FQ[b_][u_] = qrow[b_ * 16 * (d / 4) + ((C) >> 1) * 16 + qlow[(C) & 1][u_]];                 \
    }                                                                                             \
  }
#if defined(AMDR_ABLATE) && AMDR_ABLATE == 1  // timing-only build: matrix pipe removed, operands kept live
#define AMDR_MFMA_STEP(A_, B_, BI, BJ) acc[BI][BJ].x += (A_) * (B_);
#else
#define AMDR_MFMA_STEP(A_, B_, BI, BJ) acc[BI][BJ] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, acc[BI][BJ], 0, 0, 0);
#endif
// 8 k-steps x 4 accumulator blocks = 32 MFMAs of 32 cycles per chunk; an accumulator is touched
// every fourth instruction, well past the 40-cycle dependent latency.
// A operand = streamed chunk rows, B operand = queries: a 16x16 result block then has the QUERY on the lane and four
// consecutive chunk rows in the lane's registers (acc[bi][bj][r] = <row 16 bj + 4 kq + r, query 16 bi + i16>), so
// S[query][row .. row+3] is one 16-byte store (see the epilogue).
#define AMDR_MFMA_KSTEP(FX, FQ, U, COMP)                                                          \
  AMDR_MFMA_STEP(FX[0][U].COMP, FQ[0][U].COMP, 0, 0)                                              \
  AMDR_MFMA_STEP(FX[1][U].COMP, FQ[0][U].COMP, 0, 1)                                              \
  AMDR_MFMA_STEP(FX[0][U].COMP, FQ[1][U].COMP, 1, 0)                                              \
  AMDR_MFMA_STEP(FX[1][U].COMP, FQ[1][U].COMP, 1, 1)
#define AMDR_MFMA_CHUNK(FX, FQ)                                                                   \
  AMDR_MFMA_KSTEP(FX, FQ, 0, x) AMDR_MFMA_KSTEP(FX, FQ, 0, y) AMDR_MFMA_KSTEP(FX, FQ, 0, z)       \
  AMDR_MFMA_KSTEP(FX, FQ, 0, w) AMDR_MFMA_KSTEP(FX, FQ, 1, x) AMDR_MFMA_KSTEP(FX, FQ, 1, y)       \
  AMDR_MFMA_KSTEP(FX, FQ, 1, z) AMDR_MFMA_KSTEP(FX, FQ, 1, w)
// (Pinning an MFMA / ds_write / global_load / ds_read interleave with sched_group_barrier was
// measured 2-7 % SLOWER than hipcc's own schedule of this region and is not kept.)
#define AMDR_INTERLEAVE()

#if defined(AMDR_ABLATE) && AMDR_ABLATE == 2  // timing-only build: no X traffic (registers filled from an address hash)
#define AMDR_LDX(PTR) ([&] { v4f z_; z_.x = z_.y = z_.z = z_.w = (float)(((size_t)(PTR)) & 1023) * 1e-3f; return z_; }())
#else
// NTL (template constant): non-temporal policy for a matrix beyond the Infinity Cache (common.hpp)
#define AMDR_LDX(PTR) (NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(PTR)) : *reinterpret_cast<const v4f*>(PTR))
#endif

// grid: (x = row slabs, y = 32-query tiles).
// LDS: Q tile row-major [32][d/4] float4, slots XOR-swizzled per row (d*128 B) + per wave one private 4-KiB chunk stage.
// (Round 1 also ran this kernel with the roles swapped — chunk tiles in LDS, queries streamed — for
// short corpora under long batches; dense_panel.hip replaced that orientation and it was removed.)
template <int D8, int WAVES, bool NTL>  // NTL: non-temporal loads of the streamed operand; D8 = d / 8; WAVES = 8 when the Q tile leaves room for 8 stages (d <= 768)
__global__ __launch_bounds__(WAVES * 64) void dense_mfma_scores_kernel(const float* __restrict__ X, long n,
                                                                 const float* __restrict__ Q, int nq,
                                                                 long rows_per_block, int gx, int gy, long ldS,
                                                                 float* __restrict__ S /*[queries][ldS]*/, int mode,
                                                                 const int* __restrict__ tile_list,
                                                                 const int* __restrict__ tile_count, long n_real,
                                                                 const int* __restrict__ gate, int list_stride) {
  if (gate != nullptr && *gate == 0) return;  // a gated launch (dense_hi.hip): decided on the device, block-uniform
  // mode 0: S[query][row] for every row.
  // mode 1: S[query][tile] = MAXIMUM of the query's scores over the 32-row tile (first pass of the two-level
  //         top-k of a large scan: dense.hip run_search_two_level).
  // mode 2: `n` counts VIRTUAL rows, 32 per entry of tile_list; virtual tile t reads chunk tile tile_list[t]
  //         (t < *tile_count, else it is filled with -FLT_MAX) and writes S[query][32 t ..]: the exact re-scoring of
  //         the candidate tiles — same loads, same MFMA k order, the same bits as mode 0.
  // mode 3: mode 2 PER QUERY: block row `by` is query `by` alone (the LDS tile holds it in row 0, zeros below), its
  //         candidate tiles are tile_list[by * list_stride ..], tile_count[by] of them, its scores S[by][32 t ..]:
  //         a query is scored against its own candidates only, not against the union of a whole batch's.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int d = D8 * 8;
  constexpr int NCH = d / kKC;  // chunks per row: 12 / 24 / 32 (always even: d % 64 == 0)
  static_assert(d % kKC == 0, "dim must be a multiple of 64");
  const v4f* qsv = reinterpret_cast<const v4f*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* stage = smem + (size_t)d * 128 + (size_t)wave * kStageBufs * kStageBytes;
  const int i16 = lane & 15, kq = lane >> 4;  // MFMA fragment roles
  // Lane-dependent parts of the fragment addresses, computed once: every other term is a
  // compile-time constant of the unrolled chunk loop and folds into the ds_read offset field.
  //   stage: row 16 b + i16, slot 4 u + kq -> b * 2048 + i16 * 128 + ((4u + kq) ^ (i16 >> 1)) * 16
  //   LDS tile: row 16 b + i16, slot 8 C + 4 u + kq, XOR i16 touches the low four bits only
  int xoff[2], qlow[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    xoff[u] = stage_off(i16, 4 * u + kq);
    qlow[0][u] = (4 * u + kq) ^ i16;
    qlow[1][u] = (8 + 4 * u + kq) ^ i16;
  }
  const v4f* qrow = qsv + i16 * (d / 4);
  // Block id -> (slab bx of the streamed operand, tile by of the LDS operand).  The grid is 1-D and
  // remapped so that each XCD (blocks with equal id % 8 share one; 8 XCDs, each with its own L2)
  // owns a CONTIGUOUS range of the slab-major order: the `gy` blocks that stream the same slab
  // then sit on one XCD and the slab crosses the fabric once, not once per XCD.  (UCC-en step,
  // PMC: L2-miss reads of this launch 268 MB -> see profiles/; the bijective form of the remap
  // handles grids that are not a multiple of 8.)
  int bx, by;
  {
    const int nwg = gx * gy, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    // with a single LDS tile (gy == 1) nothing is shared between blocks: plain order (a 10 M-row
    // scan measured 1 % slower remapped, the UCC-en launch 1.3 % faster and 3.7x less fabric traffic)
    // (mode 3: plain order too — low bx = the head of every query's list = the live blocks; in XCD-contiguous order they
    // would all land on the first XCDs and the blocks beyond the lists on the last)
    const int logical = (gy == 1 || mode == 3) ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    bx = logical / gy;
    by = logical - bx * gy;
  }
  int q0 = by * 32;
  if (mode == 3) {
    q0 = by;
    nq = by + 1;
    tile_list += (size_t)by * list_stride;
    tile_count += by;
    // a block whose tiles all lie beyond the query's list has nothing to score: its columns can never win
    const long lo3 = (long)bx * rows_per_block;
    if ((lo3 >> 5) >= *tile_count) {
      long hi3 = lo3 + rows_per_block;
      if (hi3 > n) hi3 = n;
      for (long c = lo3 + threadIdx.x; c < hi3; c += WAVES * 64) S[(size_t)by * ldS + c] = -FLT_MAX;
      return;
    }
  }

  // ---- stage the query tile, row-major with a per-row XOR swizzle of the 16-byte slots:
  //   qs[i * d/4 + (k4 ^ (i & 15))] = Q[q0+i][4*k4 .. 4*k4+3]
  // Global reads are fully coalesced (a wave reads 1 KiB of one query row per instruction);
  // both the staging writes (8 consecutive slots of one row per ds_write_b128 lane group) and
  // the fragment reads (same logical slot of 16 different rows per ds_read_b128 lane group;
  // row stride d*4 is a multiple of the 256-B bank row) are conflict-free.
  // Loads are issued in batches of 16 per thread BEFORE any of them is consumed: one load per
  // loop trip serialises 24+ L2 round trips (measured: ~14 us of a 25-us block at UCC-en size).
  {
    constexpr int NT = WAVES * 64, TOTAL = 32 * (d / 4), PER = (TOTAL + NT - 1) / NT, BATCH = 16;
#pragma unroll
    for (int j0 = 0; j0 < PER; j0 += BATCH) {
      v4f tmp[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        if (j0 + j < PER) {
          const int idx = threadIdx.x + (j0 + j) * NT;
          const int qi_ = idx / (d / 4), k4 = idx - qi_ * (d / 4);
          v4f z = {0.f, 0.f, 0.f, 0.f};
          tmp[j] = (idx < TOTAL && q0 + qi_ < nq) ? *reinterpret_cast<const v4f*>(Q + (size_t)(q0 + qi_) * d + 4 * k4)
                                                  : z;
        }
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        if (j0 + j < PER) {
          const int idx = threadIdx.x + (j0 + j) * NT;
          const int qi_ = idx / (d / 4), k4 = idx - qi_ * (d / 4);
          if (idx < TOTAL) reinterpret_cast<v4f*>(smem)[qi_ * (d / 4) + (k4 ^ (qi_ & 15))] = tmp[j];
        }
      }
    }
  }
  __syncthreads();

  const long row_lo = (long)bx * rows_per_block;
  long row_hi = row_lo + rows_per_block;
  if (row_hi > n) row_hi = n;
  // loader role of this lane inside a 1-KiB piece: 8 rows x 8 slots
  const int lrow = lane >> 3, lslot = lane & 7;

  const int n_list = (mode >= 2) ? *tile_count : 0;
  for (long r0 = row_lo + (long)wave * 32; r0 < row_hi; r0 += (long)WAVES * 32) {
    long src0 = r0, src_hi = row_hi;  // rows actually read
    if (mode >= 2) {
      const int t = (int)(r0 >> 5);
      if (t >= n_list) {  // beyond the candidate list: columns that can never win
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
          const int q = q0 + 16 * bi + i16;
          if (q < nq) {
            float* srow = S + (size_t)q * ldS + r0 + 4 * kq;
#pragma unroll
            for (int bj = 0; bj < 2; ++bj) *reinterpret_cast<f32x4*>(srow + 16 * bj) = f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
          }
        }
        continue;
      }
      src0 = (long)tile_list[t] * 32;
      src_hi = n_real;
    }
    // global pointers of the 8 pieces (rows 4p + lrow), clamped at the slab end
    const float* gp[kPieces];
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      long r = src0 + 8 * p + lrow;
      if (r >= src_hi) r = src_hi - 1;
      gp[p] = X + (size_t)r * d + lslot * 4;
    }
    // Software pipeline; every index below is a compile-time constant once the chunk loop is
    // unrolled, so the arrays live in registers:
    //   G[c % kDepth]  chunk c as it arrives from HBM (kDepth chunks = 16 KiB per wave in flight)
    //   stage[c & 1]   this wave's LDS image of chunk c (swizzled, double-buffered)
    //   FX/FQ[c & 1]   MFMA fragments of chunk c, read from LDS one chunk AHEAD of their use so
    //                  the dependent MFMA chain never waits on an LDS round trip.
    v4f G[kDepth][kPieces], FX[2][2][2], FQ[2][2][2];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) {
      if (j < NCH) {
#pragma unroll
        for (int p = 0; p < kPieces; ++p) G[j][p] = AMDR_LDX(gp[p] + j * kKC);
      }
    }
    AMDR_STAGE_CHUNK(stage, G[0])
    if (kDepth < NCH) {
#pragma unroll
      for (int p = 0; p < kPieces; ++p) G[0][p] = AMDR_LDX(gp[p] + kDepth * kKC);
    }
    wave_lds_fence();
    AMDR_READ_FRAGS(stage, 0, FX[0], FQ[0])
    f32x4 acc[2][2];  // acc[bi][bj][r] = <streamed row 16 bj + 4 kq + r, query (LDS-tile row) 16 bi + i16>
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // One scheduling region per chunk: the 32 MFMAs of chunk c and, in their shadow,
      // the staging of chunk c+1 (4 ds_write), the refill of its registers with chunk
      // c+1+kDepth (4 loads) and its fragment reads (8 ds_read).  The LDS unit serves a wave's
      // operations in order and the compiler keeps the may-alias write->read order on `stage`.
      if (c + 1 < NCH) {
        unsigned char* st_n = stage + ((c + 1) % kStageBufs) * kStageBytes;
        AMDR_STAGE_CHUNK(st_n, G[(c + 1) % kDepth])
        if (c + 1 + kDepth < NCH) {
#pragma unroll
          for (int p = 0; p < kPieces; ++p) G[(c + 1) % kDepth][p] = AMDR_LDX(gp[p] + (c + 1 + kDepth) * kKC);
        }
        AMDR_READ_FRAGS(st_n, c + 1, FX[(c + 1) & 1], FQ[(c + 1) & 1])
      }
      AMDR_MFMA_CHUNK(FX[c & 1], FQ[c & 1])
      AMDR_INTERLEAVE()
      wave_lds_fence();
    }
    // Epilogue: four 16-byte stores per lane, S[query 16 bi + i16][rows 16 bj + 4 kq .. + 3] — a store instruction
    // covers 64 contiguous bytes of each of 16 query rows.  (Round 1 had the operands the other way round — query
    // on the registers — and turned the 32x32 tile through the wave's LDS stage to store whole 128-byte rows:
    // 16 ds_write + 16 ds_read + 16 four-byte stores per tile; with the matrix pipe taken out that kernel ran no
    // faster on the 10 M-row scan, i.e. the non-MFMA stream bound it, and 8 / 16 / 32 queries per scan cost
    // 4.9 / 5.4 / 5.8 ms.  With these stores: 4.9 / 5.1 / 5.8-5.9 ms, the UCC-en launch of this kernel 351 -> 344 us.
    // Timing-only builds on the 10 M-row, 32-query scan: without the score stores 5.80 -> 5.10 ms (1.28 GB of writes
    // = 4 % of the bytes cost 12 % of the time: they interleave with the read stream at the HBM), non-temporal
    // stores 6.01 -> 6.10 ms.)  Rows past n inside the last 32-row tile repeat row n - 1 and land in the
    // padding of S (ldS is a multiple of 32); slab boundaries are multiples of 32 rows.
    if (mode == 1) {
      // tile maxima: 7 v_max per query block in the lane, two exchanges over the four kq groups (rows past the end
      // of the matrix repeat its last row: no effect on a maximum)
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        float m = fmaxf(fmaxf(fmaxf(acc[bi][0][0], acc[bi][0][1]), fmaxf(acc[bi][0][2], acc[bi][0][3])),
                        fmaxf(fmaxf(acc[bi][1][0], acc[bi][1][1]), fmaxf(acc[bi][1][2], acc[bi][1][3])));
        m = fmaxf(m, __uint_as_float(lane_xor<16>(__float_as_uint(m))));
        m = fmaxf(m, __uint_as_float(lane_xor<32>(__float_as_uint(m))));
        const int q = q0 + 16 * bi + i16;
        if (kq == 0 && q < nq) S[(size_t)q * ldS + (r0 >> 5)] = m;
      }
      continue;
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int q = q0 + 16 * bi + i16;
      if (q < nq) {
        float* srow = S + (size_t)q * ldS + r0 + 4 * kq;
#if defined(AMDR_ABLATE) && AMDR_ABLATE == 3  // timing-only build: no score stores
        if (acc[bi][0][0] == 12345.f) srow[0] = acc[bi][1][1];
#else
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          f32x4 v = acc[bi][bj];
          if (mode >= 2) {  // rows past the end of the matrix inside the last tile are no candidates
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (src0 + 16 * bj + 4 * kq + r >= n_real) v[r] = -FLT_MAX;
          }
          *reinterpret_cast<f32x4*>(srow + 16 * bj) = v;
        }
#endif
      }
    }
  }
}


// grid: (x = row slabs, y = queries): top-k of S[q][slab] -> part[slab][q][k], or, when there
// is a single slab, straight to the final (scores, ids).  WAVES = 1 for short rows.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void scores_slab_topk_kernel(const float* __restrict__ S, long ldS, long n, int nq, int k,
                                                                       int cap, long rows_per_slab,
                                                                       C32* __restrict__ part,
                                                                       float* __restrict__ fin_scores,
                                                                       long long* __restrict__ fin_ids,
                                                                       const int* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (gate != nullptr && *gate == 0) return;
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)WAVES * cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.y;
  const long lo = (long)blockIdx.x * rows_per_slab;
  long hi = lo + rows_per_slab;
  if (hi > n) hi = n;
  const float* row = S + (size_t)qi * ldS;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  bool done = false;
  if (WAVES == 1 && k <= 64 && hi - lo <= kSelectRowsMax) {
    // short row: register selector (topk.hpp); scratch = the upper half of the staging buffer.
    // V = keys per lane, sized to the row (a UCC-en row of 591 scores needs 10, a
    // Civil-Code-zh row of 1 260 needs 20).
    int got;
    if (hi - lo <= 256)
      got = select_row<4>(row, lo, hi, k, lane, tk.buf);
    else if (hi - lo <= 640)
      got = select_row<10>(row, lo, hi, k, lane, tk.buf);
    else if (hi - lo <= 1024)
      got = select_row<16>(row, lo, hi, k, lane, tk.buf);
    else if (hi - lo <= 1280)
      got = select_row<20>(row, lo, hi, k, lane, tk.buf);
    else
      got = select_row<32>(row, lo, hi, k, lane, tk.buf);
    if (got >= 0) {
      tk.cnt = got;
      done = true;
    }
  }
  if (!done) {
    // S is read exactly once: 16-byte non-temporal loads, four consecutive rows per lane (slabs start
    // on multiples of 64 rows and S rows on 128-byte lines, so every float4 below `hi` rounded up to 4
    // lies inside the padded row)
    for (long base = lo + (long)wave * 256; base < hi; base += (long)WAVES * 256) {
      const long r0 = base + 4 * lane;
      const v4f z = {0.f, 0.f, 0.f, 0.f};
      const v4f x = (r0 < hi) ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(row + r0)) : z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long r = r0 + e;
        const bool v = r < hi;
        tk.push_lanes(v ? C32::make(x[e], (u32)r) : C32::pad(), v, lane);
      }
    }
    tk.finalize(lane);
  }
  if (WAVES > 1) block_combine_topk(tk, lists, cap, WAVES, wave, lane, cnts);
  if (wave == 0) {
    if (fin_ids) {
      for (int j = lane; j < k; j += 64) {
        const bool v = j < tk.cnt;
        const C32 c = v ? tk.buf[j] : C32::pad();
        fin_scores[(size_t)qi * k + j] = v ? c.score() : -FLT_MAX;
        fin_ids[(size_t)qi * k + j] = v ? c.id() : -1ll;
      }
    } else {
      C32* dst = part + ((size_t)blockIdx.x * nq + qi) * k;
      for (int j = lane; j < k; j += 64) dst[j] = (j < tk.cnt) ? tk.buf[j] : C32::pad();
    }
  }
}

__global__ __launch_bounds__(64) void scores_pair_topk_kernel(const float* __restrict__ S, long ldS, long n, int nq,
                                                              int k, int cap, float* __restrict__ fin_scores,
                                                              long long* __restrict__ fin_ids,
                                                              const int* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (gate != nullptr && *gate == 0) return;
  C32* buf = reinterpret_cast<C32*>(smem);  // cap entries (>= 128): staged-selector list; the pair selector uses 64
  const int lane = threadIdx.x;
  const int q = 2 * blockIdx.x + (lane >> 5);
  const bool has_q = q < nq;
  C32 out = C32::pad();
  const int got = select_row_pair_any(S, ldS, n, q, has_q, k, lane, buf, out);
  if (got >= 0) {
    const int j = lane & 31;
    if (has_q && j < k) {
      const bool v = j < got;
      fin_scores[(size_t)q * k + j] = v ? out.score() : -FLT_MAX;
      fin_ids[(size_t)q * k + j] = v ? out.id() : -1ll;
    }
    return;
  }
  // mass ties at the cut in one of the two rows: the general selector, one row after the other
  for (int h = 0; h < 2; ++h) {
    const int qq = 2 * blockIdx.x + h;
    if (qq >= nq) break;
    const float* row = S + (size_t)qq * ldS;
    WaveTopK<C32> tk;
    tk.init(buf, cap, k);
    for (long base = 0; base < n; base += 64) {
      const long r = base + lane;
      const bool v = r < n;
      tk.push_lanes(v ? C32::make(row[r], (u32)r) : C32::pad(), v, lane);
    }
    tk.finalize(lane);
    for (int j = lane; j < k; j += 64) {
      const bool v = j < tk.cnt;
      const C32 c = v ? tk.buf[j] : C32::pad();
      fin_scores[(size_t)qq * k + j] = v ? c.score() : -FLT_MAX;
      fin_ids[(size_t)qq * k + j] = v ? c.id() : -1ll;
    }
    wave_lds_fence();
  }
}

// Two-level top-k of a large scan, step 2: the <= 8 192 candidate tile ids (k per query, -1 = none) -> ascending
// list without duplicates + its length.  One wave; bitonic sort in LDS (descending on id + 1, so that "none" sorts
// last), neighbour compare, one prefix sum.
__global__ __launch_bounds__(64) void tiles_unique_kernel(const long long* __restrict__ tile_ids, int n_in, int cap,
                                                          int* __restrict__ list, int* __restrict__ count,
                                                          const int* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (gate != nullptr && *gate == 0) return;
  C32* buf = reinterpret_cast<C32*>(smem);
  const int lane = threadIdx.x;
  // block b sorts ids [b * n_in, (b + 1) * n_in) into list + b * n_in, count[b]: one block for the union of a batch,
  // one per query for the per-query candidate lists behind the fp16 first pass
  tile_ids += (size_t)blockIdx.x * n_in;
  list += (size_t)blockIdx.x * n_in;
  count += blockIdx.x;
  for (int i = lane; i < cap; i += 64) {
    const long long v = i < n_in ? tile_ids[i] : -1ll;
    buf[i].c = v >= 0 ? (u64)(v + 1) : 0ull;
  }
  wave_bitonic_sort_desc(buf, cap, lane);
  const int per = cap / 64;  // cap is a power of two >= 64
  int mine = 0;
  for (int j = 0; j < per; ++j) {
    const int i = lane * per + j;
    const u64 x = buf[i].c, prev = i ? buf[i - 1].c : ~0ull;
    mine += (x != 0ull && x != prev) ? 1 : 0;
  }
  int incl = mine;
#pragma unroll
  for (int s2 = 1; s2 < 64; s2 <<= 1) {
    const int o = __shfl_up(incl, s2);
    incl += (lane >= s2) ? o : 0;
  }
  const int total = __builtin_amdgcn_readlane(incl, 63);
  int pos = incl - mine;
  for (int j = 0; j < per; ++j) {
    const int i = lane * per + j;
    const u64 x = buf[i].c, prev = i ? buf[i - 1].c : ~0ull;
    if (x != 0ull && x != prev) list[total - 1 - pos++] = (int)(x - 1);  // descending order in, ascending out
  }
  if (lane == 0) *count = total;
}

// The same list from a bitmap of the tiles in LDS (one block of 1 024 threads, up to 2^20 tiles = 128 KiB of bits):
// mark, popcount prefix sum over the words, emit ascending.  The one-wave sort above takes 0.3 ms for the 2 112
// candidates of a 64-query pass behind the fp16 first pass — more than every other step after the scan together.
__global__ __launch_bounds__(1024) void tiles_unique_bitmap_kernel(const long long* __restrict__ tile_ids, int n_in,
                                                                   int n_tiles, int* __restrict__ list,
                                                                   int* __restrict__ count,
                                                                   const int* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (gate != nullptr && *gate == 0) return;
  const int words = (n_tiles + 31) >> 5;
  unsigned int* bm = reinterpret_cast<unsigned int*>(smem);
  int* wsum = reinterpret_cast<int*>(bm + words);  // 16 wave totals
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < words; i += 1024) bm[i] = 0u;
  __syncthreads();
  for (int i = tid; i < n_in; i += 1024) {
    const long long v = tile_ids[i];
    if (v >= 0 && v < n_tiles) atomicOr(&bm[v >> 5], 1u << (v & 31));
  }
  __syncthreads();
  const int per = (words + 1023) / 1024, w0 = tid * per;
  int mine = 0;
  for (int j = 0; j < per; ++j)
    if (w0 + j < words) mine += __popc(bm[w0 + j]);
  int incl = mine;
#pragma unroll
  for (int s2 = 1; s2 < 64; s2 <<= 1) {
    const int o = __shfl_up(incl, s2);
    incl += (lane >= s2) ? o : 0;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    const int x = wsum[w];
    base += w < wave ? x : 0;
    total += x;
  }
  int pos = base + incl - mine;
  for (int j = 0; j < per; ++j) {
    if (w0 + j >= words) break;
    unsigned int x = bm[w0 + j];
    while (x) {
      const int b = __ffs((int)x) - 1;
      list[pos++] = (w0 + j) * 32 + b;
      x &= x - 1;
    }
  }
  if (tid == 0) *count = total;
}

// step 4: the final hits carry COLUMNS of the re-scored candidate matrix (32 per list entry); columns ascend with the
// row ids (the list is ascending), so ties were already broken towards the lower id.
__global__ __launch_bounds__(256) void tiles_remap_ids_kernel(long long* __restrict__ ids, int total,
                                                              const int* __restrict__ list,
                                                              const int* __restrict__ count, long n_real,
                                                              const int* __restrict__ gate, int k, int list_stride) {
  if (gate != nullptr && *gate == 0) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < total) {
    if (list_stride > 0) {  // per-query lists: entry i belongs to query i / k
      list += (size_t)(i / k) * list_stride;
      count += i / k;
    }
    const long long c = ids[i];
    if (c >= 0) {
      // a filler column (beyond the candidate list, or past the end of the matrix inside the last tile: score
      // -FLT_MAX) reaches the final list only when fewer than k candidate rows have a real (non-NaN) score: it is
      // padding, id -1 (the faiss convention for "fewer than k results"), never a read of an unwritten list entry
      const long long t = c >> 5;
      const long long r = t < *count ? (long long)list[t] * 32 + (c & 31) : -1;
      ids[i] = r < n_real ? r : -1;
    }
  }
}

int dense_tiles_unique_launch(const int64_t* tile_ids, int n_in, long n_tiles, int* list, int* count, hipStream_t st,
                              const int* gate) {
  if (n_tiles > 0 && n_tiles <= kUniqueBitmapTilesMax && n_in >= 512) {
    const size_t lds = (size_t)((n_tiles + 31) / 32) * 4 + 64;
    AMDR_HIP(hipFuncSetAttribute((const void*)tiles_unique_bitmap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 128 * 1024 + 64));
    hipLaunchKernelGGL(tiles_unique_bitmap_kernel, dim3(1), dim3(1024), lds, st, (const long long*)tile_ids, n_in,
                       (int)n_tiles, list, count, gate);
    AMDR_HIP(hipGetLastError());
    return AMDR_OK;
  }
  int cap = 64;
  while (cap < n_in) cap <<= 1;
  hipLaunchKernelGGL(tiles_unique_kernel, dim3(1), dim3(64), (size_t)cap * sizeof(C32), st, (const long long*)tile_ids, n_in,
                     cap, list, count, gate);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}
// one sorted duplicate-free list per query: ids [m][kc] -> list [m][kc] (ascending), count [m]
int dense_tiles_sort_per_query_launch(const int64_t* tile_ids, int m, int kc, int* list, int* count, hipStream_t st) {
  int cap = 64;
  while (cap < kc) cap <<= 1;
  hipLaunchKernelGGL(tiles_unique_kernel, dim3(m), dim3(64), (size_t)cap * sizeof(C32), st, (const long long*)tile_ids, kc, cap,
                     list, count, (const int*)nullptr);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}
int dense_tiles_remap_launch(int64_t* ids, int total, const int* list, const int* count, long n_real, hipStream_t st,
                             const int* gate, int k, int list_stride) {
  hipLaunchKernelGGL(tiles_remap_ids_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (long long*)ids, total, list,
                     count, n_real, gate, k, list_stride);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: exact scores of every query's OWN candidate tiles behind the fp16 first pass, lean form.  Mode 3 of the tile
// kernel above did this with a whole 32-query LDS tile per block (96 KiB staged, 31 of 32 MFMA columns idle, one block
// per CU: 38 us per 64 queries).  Here a wave scores ONE (query, tile): the tile's rows stream through the wave's 4-KiB
// stage exactly as above (same loads, same fragments), the B operand is the query's own components broadcast to all 16
// columns, and only the two row blocks are multiplied: 16 MFMAs per chunk instead of 32, the k-steps in the SAME order —
// acc[bj] sees the sequence the tile kernel's acc[bi][bj] sees, hence the same bits (tested against modes 0 / 2 / 3).
// 20 KiB of LDS per 4-wave block: the candidate tiles of a batch spread over every CU.
// grid: (queries, ceil(max tiles per query / WPB)).
template <int D8, int WPB>
__global__ __launch_bounds__(WPB * 64) void dense_rescore_tiles_kernel(const float* __restrict__ X, long n_real,
                                                                       const float* __restrict__ Q,
                                                                       const int* __restrict__ list,
                                                                       const int* __restrict__ count, int list_stride,
                                                                       long ldS, float* __restrict__ S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int d = D8 * 8;
  constexpr int NCH = d / kKC;
  // 8 chunks (32 KiB) in flight per wave: a wave walks ONE tile, 24-32 dependent chunk steps, and with 4 in flight the kernel
  // was bound by the latency of its own loads (64 queries: 27 us for 86 MB)
  constexpr int kRsDepth = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // grid (queries, tile groups): the blocks that hold the HEADS of the lists — the live ones — are dispatched first and
  // land one per CU; with the tile group as the fast index the 4 live blocks of a query sat next to 5 dead ones and the
  // live blocks clumped on the CUs the dead ones had just freed (27 us for 86 MB)
  const int q = blockIdx.x;
  const int cnt = count[q];
  if ((int)blockIdx.y * WPB >= cnt) return;  // block-uniform: nothing of this query's list falls to the block
  v4f* qv = reinterpret_cast<v4f*>(smem);
  for (int i = threadIdx.x; i < d / 4; i += WPB * 64) qv[i] = *reinterpret_cast<const v4f*>(Q + (size_t)q * d + 4 * i);
  __syncthreads();
  const int t = blockIdx.y * WPB + wave;
  if (t >= cnt) return;
  unsigned char* stage = smem + (size_t)d * 4 + (size_t)wave * kStageBytes;
  const int i16 = lane & 15, kq = lane >> 4;
  const int lrow = lane >> 3, lslot = lane & 7;
  int xoff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) xoff[u] = stage_off(i16, 4 * u + kq);
  const long src0 = (long)list[(size_t)q * list_stride + t] * 32;
  const float* gp[kPieces];
#pragma unroll
  for (int p = 0; p < kPieces; ++p) {
    long r = src0 + 8 * p + lrow;
    if (r >= n_real) r = n_real - 1;
    gp[p] = X + (size_t)r * d + lslot * 4;
  }
  v4f G[kRsDepth][kPieces], FX[2][2][2], FQ[2][2];
#pragma unroll
  for (int j = 0; j < kRsDepth; ++j) {
    if (j < NCH) {
#pragma unroll
      for (int p = 0; p < kPieces; ++p) G[j][p] = *reinterpret_cast<const v4f*>(gp[p] + j * kKC);
    }
  }
#define AMDR_RS_FRAGS(ST, C, FX_, FQ_)                                                     \
  _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                       \
    FX_[0][u_] = *reinterpret_cast<const v4f*>((ST) + xoff[u_]);                           \
    FX_[1][u_] = *reinterpret_cast<const v4f*>((ST) + 2048 + xoff[u_]);                    \
    FQ_[u_] = qv[8 * (C) + 4 * u_ + kq];                                                   \
  }
#define AMDR_RS_KSTEP(FX_, FQ_, U, COMP)                                                              \
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(FX_[0][U].COMP, FQ_[U].COMP, acc[0], 0, 0, 0);        \
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(FX_[1][U].COMP, FQ_[U].COMP, acc[1], 0, 0, 0);
  AMDR_STAGE_CHUNK(stage, G[0])
  if (kRsDepth < NCH) {
#pragma unroll
    for (int p = 0; p < kPieces; ++p) G[0][p] = *reinterpret_cast<const v4f*>(gp[p] + kRsDepth * kKC);
  }
  wave_lds_fence();
  AMDR_RS_FRAGS(stage, 0, FX[0], FQ[0])
  f32x4 acc[2];  // acc[bj][r] = <row 16 bj + 4 kq + r of the tile, the query>, the same in every column i16
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) {
      AMDR_STAGE_CHUNK(stage, G[(c + 1) % kRsDepth])
      if (c + 1 + kRsDepth < NCH) {
#pragma unroll
        for (int p = 0; p < kPieces; ++p) G[(c + 1) % kRsDepth][p] = *reinterpret_cast<const v4f*>(gp[p] + (c + 1 + kRsDepth) * kKC);
      }
      AMDR_RS_FRAGS(stage, c + 1, FX[(c + 1) & 1], FQ[(c + 1) & 1])
    }
    AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 0, x) AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 0, y)
    AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 0, z) AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 0, w)
    AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 1, x) AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 1, y)
    AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 1, z) AMDR_RS_KSTEP(FX[c & 1], FQ[c & 1], 1, w)
    wave_lds_fence();
  }
#undef AMDR_RS_FRAGS
#undef AMDR_RS_KSTEP
  if (i16 == 0) {
    float* srow = S + (size_t)q * ldS + (size_t)t * 32 + 4 * kq;
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      f32x4 v = acc[bj];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (src0 + 16 * bj + 4 * kq + r >= n_real) v[r] = -FLT_MAX;  // rows past the end inside the last tile
      *reinterpret_cast<f32x4*>(srow + 16 * bj) = v;
    }
  }
}

// The final top-k of every query over the re-scored columns of ITS tiles (32 per list entry, count[q] entries), with the
// column -> row id map in the epilogue (columns ascend with the rows: ties already went to the lower id).  A column
// past the end of the matrix (score -FLT_MAX) can only surface when fewer than k rows are real: id -1.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dense_final_topk_kernel(const float* __restrict__ S, long ldS,
                                                                      const int* __restrict__ list,
                                                                      const int* __restrict__ count, int list_stride,
                                                                      long n_real, int k, int cap,
                                                                      float* __restrict__ fin_scores,
                                                                      long long* __restrict__ fin_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)WAVES * cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x;
  const long hi = (long)count[qi] * 32;
  const float* row = S + (size_t)qi * ldS;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  bool done = false;
  if (WAVES == 1 && k <= 64 && hi <= kSelectRowsMax) {
    int got;
    if (hi <= 512)
      got = select_row<8>(row, 0, hi, k, lane, tk.buf);
    else if (hi <= 1024)
      got = select_row<16>(row, 0, hi, k, lane, tk.buf);
    else
      got = select_row<32>(row, 0, hi, k, lane, tk.buf);
    if (got >= 0) {
      tk.cnt = got;
      done = true;
    }
  }
  if (!done) {
    for (long base = (long)wave * 256; base < hi; base += (long)WAVES * 256) {
      const long r0 = base + 4 * lane;
      const v4f z = {0.f, 0.f, 0.f, 0.f};
      const v4f x = (r0 < hi) ? *reinterpret_cast<const v4f*>(row + r0) : z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long r = r0 + e;
        const bool v = r < hi;
        tk.push_lanes(v ? C32::make(x[e], (u32)r) : C32::pad(), v, lane);
      }
    }
    tk.finalize(lane);
  }
  if (WAVES > 1) block_combine_topk(tk, lists, cap, WAVES, wave, lane, cnts);
  if (wave == 0) {
    for (int j = lane; j < k; j += 64) {
      // A NaN score (key 1: it sorts behind every real score) is no hit: the exact two-level form ranks its filler
      // columns (-FLT_MAX, id -1) above NaN rows, so a query or rows that score NaN come back as padding there — the
      // same here (tests/test_dense_hi_gpu.py: the exact two-level form decides what a NaN query returns).
      const bool v = j < tk.cnt && (u32)(tk.buf[j].c >> 32) != 1u;
      const C32 c = v ? tk.buf[j] : C32::pad();
      long long id = -1ll;
      if (v) {
        const long long col = c.id();
        const long long r = (long long)list[(size_t)qi * list_stride + (col >> 5)] * 32 + (col & 31);
        id = r < n_real ? r : -1ll;
      }
      fin_scores[(size_t)qi * k + j] = v ? c.score() : -FLT_MAX;
      fin_ids[(size_t)qi * k + j] = id;
    }
  }
}

bool dense_mfma_supported(int d) { return d >= 64 && d <= 1024 && d % 64 == 0; }

// Plan shared by reserve and launch.
// waves per block: each needs kStageBufs x 4 KiB of stage beside the d*128-byte LDS tile in 160 KiB of LDS
static constexpr int scores_waves(int d) {
  int w = (int)((160 * 1024 - (long)d * 128) / (kStageBufs * kStageBytes));
  return w > 8 ? 8 : w;  // two waves per SIMD wherever the LDS tile leaves room (every d <= 1024 with one stage)
}

// `tiles` 32-row tiles of the chunk matrix are cut into gx slabs per query tile held in LDS
// (`q_tiles` of them).  One block per CU is resident (the LDS tile fills most of the LDS), a block
// costs (tiles per wave) tile-times plus ~1/4 tile-time to stage its LDS tile, and blocks run in
// rounds of 256.  Returns the slab count with the smallest estimate.
static long plan_slabs(long tiles, long q_tiles, int waves) {
  long gx_max = (tiles + waves - 1) / waves;  // at least one tile per wave
  if (gx_max < 1) gx_max = 1;
  long gx = 1;
  double best = 1e300;
  for (long g = 1; g <= gx_max && g <= 4096; ++g) {
    long tpb = (tiles + g - 1) / g;
    long per_wave = (tpb + waves - 1) / waves;
    long rounds = (q_tiles * g + 255) / 256;
    double est = (double)rounds * ((double)per_wave + 0.25);
    if (est < best - 1e-9) {
      best = est;
      gx = g;
    }
  }
  return gx;
}

void dense_mfma_plan(long n, int d, int nq, int k, DenseMfmaPlan* p) {
  const int kBW = scores_waves(d);
  p->q_tiles = ceil_div(nq, 32);
  const long tiles = (n + 31) / 32;
  const long gx = plan_slabs(tiles, p->q_tiles, kBW);
  long tiles_per_block = (tiles + gx - 1) / gx;
  tiles_per_block = ((tiles_per_block + kBW - 1) / kBW) * kBW;
  p->rows_per_block = tiles_per_block * 32;
  p->grid_x = (int)((n + p->rows_per_block - 1) / p->rows_per_block);
  p->grid_y = p->q_tiles;
  if (p->grid_x < 1) p->grid_x = 1;
  p->lds_scores = (size_t)d * 32 * sizeof(float) + (size_t)kBW * kStageBufs * kStageBytes;
  p->waves = kBW;
  // top-k pass: slabs of >= 16 Ki rows, enough blocks to fill the chip
  long sl = (256L * 8 + nq - 1) / nq;
  long sl_max = (n + 16383) / 16384;
  if (sl > sl_max) sl = sl_max;
  if (sl < 1) sl = 1;
  p->rows_per_slab = ((n + sl - 1) / sl + 63) / 64 * 64;
  p->slabs = (int)((n + p->rows_per_slab - 1) / p->rows_per_slab);
  if (p->slabs < 1) p->slabs = 1;
  p->cap = topk_cap(k);
  // rows of S start on 128-byte lines: the 32-column store segments are then whole lines
  p->ld = (n + 31) / 32 * 32;
  p->s_bytes = (size_t)nq * (size_t)p->ld * sizeof(float);
  p->part_bytes = (size_t)p->slabs * nq * k * sizeof(C32);
}

template <int D8, int WAVES, bool NTL>
static int launch_scores(const DenseMfmaPlan& p, const float* X, long n, const float* Q, int nq, float* S,
                         hipStream_t st, int mode, const int* tile_list, const int* tile_count, long n_real,
                         const int* gate, int list_stride) {
  // 128-160 KiB of dynamic LDS needs the opt-in.  The attribute belongs to the (function, device)
  // pair, the C ABI takes a device ordinal, and setting it is cheap: set on every launch for the
  // current device rather than remembering "done" per process.
  AMDR_HIP(hipFuncSetAttribute((const void*)dense_mfma_scores_kernel<D8, WAVES, NTL>,
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               D8 * 8 * 32 * (int)sizeof(float) + WAVES * kStageBufs * kStageBytes));
  hipLaunchKernelGGL((dense_mfma_scores_kernel<D8, WAVES, NTL>), dim3(p.grid_x * p.grid_y), dim3(WAVES * 64),
                     p.lds_scores, st, X, n, Q, nq, p.rows_per_block, p.grid_x, p.grid_y, p.ld, S, mode, tile_list, tile_count,
                     n_real, gate, list_stride);
  return AMDR_OK;
}

int dense_mfma_launch_scores(const DenseMfmaPlan& p, const float* X, long n, int d, const float* Q, int nq, float* S,
                             hipStream_t st, int mode, const int* tile_list, const int* tile_count, long n_real,
                             const int* gate, int list_stride) {
  int rc = AMDR_OK;
  const bool nt = dense_stream_nontemporal(mode >= 2 ? n_real : n, d) && mode < 2;  // candidate tiles are re-read: cacheable
  switch (d) {
#define AMDR_CASE(D)                                                          \
  case D:                                                                     \
    rc = nt ? launch_scores<D / 8, scores_waves(D), true>(p, X, n, Q, nq, S, st, mode, tile_list, tile_count, n_real, gate, list_stride)   \
            : launch_scores<D / 8, scores_waves(D), false>(p, X, n, Q, nq, S, st, mode, tile_list, tile_count, n_real, gate, list_stride); \
    break;
    AMDR_CASE(64) AMDR_CASE(128) AMDR_CASE(192) AMDR_CASE(256) AMDR_CASE(320) AMDR_CASE(384)
    AMDR_CASE(448) AMDR_CASE(512) AMDR_CASE(576) AMDR_CASE(640) AMDR_CASE(704) AMDR_CASE(768)
    AMDR_CASE(832) AMDR_CASE(896) AMDR_CASE(960) AMDR_CASE(1024)
#undef AMDR_CASE
    default: return fail(AMDR_EINVAL, "dense (batched): unsupported dim %d", d);
  }
  if (rc) return rc;
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int dense_mfma_launch_topk(const DenseMfmaPlan& p, const float* S, long n, int nq, int k, void* part,
                           float* fin_scores, int64_t* fin_ids, hipStream_t st, const int* gate) {
  const int waves = p.rows_per_slab <= kSelectRowsMax ? 1 : kBW;
  size_t lds = (size_t)waves * p.cap * sizeof(C32) + waves * sizeof(int);
  const char* pair_env = getenv("AMDR_TOPK_PAIR");  // "0" pins one query per wave (A/B, tests)
  const bool pair_off = pair_env && pair_env[0] == '0';
  if (fin_ids && p.slabs == 1 && n <= 1024 && k <= 32 && nq >= 2 && !pair_off) {
    hipLaunchKernelGGL(scores_pair_topk_kernel, dim3((nq + 1) / 2), dim3(64), (size_t)p.cap * sizeof(C32), st, S, p.ld,
                       n, nq, k, p.cap, fin_scores, (long long*)fin_ids, gate);
    AMDR_HIP(hipGetLastError());
    return AMDR_OK;
  }
  if (waves == 1)
    hipLaunchKernelGGL(scores_slab_topk_kernel<1>, dim3(p.slabs, nq), dim3(64), lds, st, S, p.ld, n, nq, k, p.cap,
                       p.rows_per_slab, (C32*)part, fin_scores, (long long*)fin_ids, gate);
  else
    hipLaunchKernelGGL(scores_slab_topk_kernel<kBW>, dim3(p.slabs, nq), dim3(256), lds, st, S, p.ld, n, nq, k, p.cap,
                       p.rows_per_slab, (C32*)part, fin_scores, (long long*)fin_ids, gate);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}


template <int D8>
static int launch_rescore(const float* X, long n_real, const float* Q, int m, const int* list, const int* count,
                          int list_stride, int max_tiles, long ldS, float* S, hipStream_t st) {
  constexpr int WPB = 4;
  const size_t lds = (size_t)D8 * 8 * sizeof(float) + (size_t)WPB * kStageBytes;
  hipLaunchKernelGGL((dense_rescore_tiles_kernel<D8, WPB>), dim3(m, (max_tiles + WPB - 1) / WPB), dim3(WPB * 64), lds, st, X,
                     n_real, Q, list, count, list_stride, ldS, S);
  return AMDR_OK;
}
// S[q][32 t ..] = exact scores of the rows of tile list[q * list_stride + t], t < count[q] (<= max_tiles)
int dense_rescore_tiles_launch(const float* X, long n_real, int d, const float* Q, int m, const int* list, const int* count,
                               int list_stride, int max_tiles, long ldS, float* S, hipStream_t st) {
  int rc = AMDR_OK;
  switch (d) {
#define AMDR_CASE(D) \
  case D: rc = launch_rescore<D / 8>(X, n_real, Q, m, list, count, list_stride, max_tiles, ldS, S, st); break;
    AMDR_CASE(64) AMDR_CASE(128) AMDR_CASE(192) AMDR_CASE(256) AMDR_CASE(320) AMDR_CASE(384)
    AMDR_CASE(448) AMDR_CASE(512) AMDR_CASE(576) AMDR_CASE(640) AMDR_CASE(704) AMDR_CASE(768)
    AMDR_CASE(832) AMDR_CASE(896) AMDR_CASE(960) AMDR_CASE(1024)
#undef AMDR_CASE
    default: return fail(AMDR_EINVAL, "dense (re-scoring): unsupported dim %d", d);
  }
  if (rc) return rc;
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}
// final (scores, row ids) of every query from S (dense_rescore_tiles_launch's layout)
int dense_final_topk_launch(const float* S, long ldS, const int* list, const int* count, int list_stride, int max_tiles,
                            long n_real, int m, int k, float* fin_scores, int64_t* fin_ids, hipStream_t st) {
  const int cap = topk_cap(k);
  const bool one = (long)max_tiles * 32 <= kSelectRowsMax && k <= 64;
  const int waves = one ? 1 : kBW;
  const size_t lds = (size_t)waves * cap * sizeof(C32) + waves * sizeof(int);
  if (one)
    hipLaunchKernelGGL(dense_final_topk_kernel<1>, dim3(m), dim3(64), lds, st, S, ldS, list, count, list_stride, n_real, k, cap,
                       fin_scores, (long long*)fin_ids);
  else
    hipLaunchKernelGGL(dense_final_topk_kernel<kBW>, dim3(m), dim3(kBW * 64), lds, st, S, ldS, list, count, list_stride, n_real,
                       k, cap, fin_scores, (long long*)fin_ids);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

}  // namespace amdr
