// Dense channel, batched form: 32 queries share one pass over the chunk matrix.
//
// Same contract as dense.hip (exact inner-product top-k behind faiss
// `index.search`, legalrag/retrieval/dense_retriever.py:42) for query batches.
// With B queries per pass the scan is a [n x d] . [d x B] product; beyond ~8
// queries the per-(row, query) cross-lane reductions of the GEMV form saturate
// the vector ALU before HBM, so the batch is tiled 32 queries wide on the
// fp32-input matrix instruction v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf
// chain (exact fp32 — no TF32/bf16 shortcut exists or is wanted), the same peak
// rate as the vector ALU, but the 32x32 accumulate needs no cross-lane traffic.
//   A (32 queries x 2): lane (i = l&31, h = l>>5) <- Q[q0+i][k]   (LDS, staged once per block)
//   B (2 x 32 rows)   : lane (j = l&31, h)        <- X[r0+j][k]   (HBM -> registers -> LDS -> registers)
// X is fetched with fully coalesced 16-B/lane loads (4 rows x 256 B per wave
// instruction, each byte of X read from HBM exactly once), parked in a wave-private,
// XOR-swizzled 8-KiB LDS stage and read back row-per-lane as the MFMA wants it; two
// chunks (16 KiB per wave) are always in flight.  The k order inside a chunk is
// permuted (lane half h takes slots 8h..8h+7); A and B use the same permutation,
// which a dot product cannot see.  C[query][row] comes
// back with the row on the lane, so each accumulator register is stored as two
// 128-byte segments of the score matrix S[query][row].  Top-k is a second,
// slab-parallel pass over S (+8 % traffic at d = 768: 128 B written and read per
// 3072-B row per 32 queries).
#include "common.hpp"
#include "topk.hpp"

#include <cfloat>

namespace amdr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBW = 4;          // waves per block
constexpr int kKC = 64;         // floats of every row per staged chunk (256 B = 16 slots of 16 B)
constexpr int kStageBytes = 32 * kKC * 4;  // 8 KiB of LDS per wave: one 32-row x 64-float chunk

// LDS image of a staged chunk: row r (0..31) at byte r*256, its logical 16-B slot s at
// physical slot s ^ (r & 15), so that the 16 lanes of a ds_read_b128 group (16 different
// rows, same logical slot) land on 16 different slots of the 256-B bank row.
__device__ __forceinline__ int stage_off(int row, int slot) { return row * 256 + ((slot ^ (row & 15)) << 4); }

// grid: (x = row slabs, y = 32-query tiles).
// LDS: Q tile as [d/4][32] float4 (d*128 B) + kBW private 8-KiB chunk stages.
template <int D8>  // D8 = d / 8
__global__ __launch_bounds__(256) void dense_mfma_scores_kernel(const float* __restrict__ X, long n,
                                                                 const float* __restrict__ Q, int nq,
                                                                 long rows_per_block,
                                                                 float* __restrict__ S /*[nq, n]*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int d = D8 * 8;
  constexpr int NCH = d / kKC;  // chunks per row: 6 / 12 / 16
  static_assert(d % kKC == 0, "dim must be a multiple of 64");
  float4* qs = reinterpret_cast<float4*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* stage = smem + (size_t)d * 128 + (size_t)wave * kStageBytes;
  const int i = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.y * 32;

  // ---- stage the query tile: qs[k4 * 32 + i] = Q[q0+i][4*k4 .. 4*k4+3], k4 in [0, d/4)
  for (int k4 = threadIdx.x >> 5; k4 < d / 4; k4 += 8) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + i < nq) v = *reinterpret_cast<const float4*>(Q + (size_t)(q0 + i) * d + 4 * k4);
    qs[k4 * 32 + i] = v;
  }
  __syncthreads();

  const long row_lo = (long)blockIdx.x * rows_per_block;
  long row_hi = row_lo + rows_per_block;
  if (row_hi > n) row_hi = n;
  // loader role of this lane inside a 1-KiB piece: 4 rows x 16 slots
  const int lrow = lane >> 4, lslot = lane & 15;

  for (long r0 = row_lo + (long)wave * 32; r0 < row_hi; r0 += (long)kBW * 32) {
    // global pointers of the 8 pieces (rows 4p + lrow), clamped at the slab end
    const float* gp[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      long r = r0 + 4 * p + lrow;
      if (r >= row_hi) r = row_hi - 1;
      gp[p] = X + (size_t)r * d + lslot * 4;
    }
    float4 g0[8], g1[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) g0[p] = *reinterpret_cast<const float4*>(gp[p]);
    if (NCH > 1) {
#pragma unroll
      for (int p = 0; p < 8; ++p) g1[p] = *reinterpret_cast<const float4*>(gp[p] + kKC);
    }
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // registers -> this wave's LDS stage (swizzled), then refill the registers two chunks ahead
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const float4 v = (c & 1) ? g1[p] : g0[p];
        *reinterpret_cast<float4*>(stage + stage_off(4 * p + lrow, lslot)) = v;
      }
      if (c + 2 < NCH) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const float4 v = *reinterpret_cast<const float4*>(gp[p] + (c + 2) * kKC);
          if (c & 1) g1[p] = v; else g0[p] = v;
        }
      }
      wave_lds_fence();
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int sl = h * 8 + m;  // logical slot of this lane's half
        const float4 xv = *reinterpret_cast<const float4*>(stage + stage_off(i, sl));
        const float4 qv = qs[(c * 16 + sl) * 32 + i];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.x, xv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.y, xv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.z, xv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.w, xv.w, acc, 0, 0, 0);
      }
      wave_lds_fence();
    }
    const long r = r0 + i;
    if (r < row_hi) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int qrow = (g & 3) + 8 * (g >> 2) + 4 * h;  // C/D map: row = query within the tile
        if (q0 + qrow < nq) S[(size_t)(q0 + qrow) * n + r] = acc[g];
      }
    }
  }
}

// grid: (x = row slabs, y = queries): top-k of S[q][slab] -> part[slab][q][k], or, when there
// is a single slab, straight to the final (scores, ids).  WAVES = 1 for short rows.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void scores_slab_topk_kernel(const float* __restrict__ S, long n, int nq, int k,
                                                                       int cap, long rows_per_slab,
                                                                       C32* __restrict__ part,
                                                                       float* __restrict__ fin_scores,
                                                                       long long* __restrict__ fin_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* lists = reinterpret_cast<C32*>(smem);
  int* cnts = reinterpret_cast<int*>(lists + (size_t)WAVES * cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.y;
  const long lo = (long)blockIdx.x * rows_per_slab;
  long hi = lo + rows_per_slab;
  if (hi > n) hi = n;
  const float* row = S + (size_t)qi * n;
  WaveTopK<C32> tk;
  tk.init(lists + (size_t)wave * cap, cap, k);
  for (long base = lo + (long)wave * 64; base < hi; base += (long)WAVES * 64) {
    const long r = base + lane;
    const bool v = r < hi;
    C32 c = v ? C32::make(row[r], (u32)r) : C32::pad();
    tk.push_lanes(c, v, lane);
  }
  tk.finalize(lane);
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

bool dense_mfma_supported(int d) { return d == 384 || d == 768 || d == 1024; }

// Plan shared by reserve and launch.
void dense_mfma_plan(long n, int d, int nq, int k, DenseMfmaPlan* p) {
  p->q_tiles = ceil_div(nq, 32);
  long tiles = (n + 31) / 32;
  long want = 256L * 2;  // one block per CU is resident (Q tile fills most of the LDS); 2 rounds balance the tail
  long gx = (want + p->q_tiles - 1) / p->q_tiles;
  long gx_max = (tiles + kBW - 1) / kBW;  // at least one tile per wave
  if (gx > gx_max) gx = gx_max;
  if (gx < 1) gx = 1;
  long tiles_per_block = (tiles + gx - 1) / gx;
  tiles_per_block = ((tiles_per_block + kBW - 1) / kBW) * kBW;
  p->rows_per_block = tiles_per_block * 32;
  p->grid_x = (int)((n + p->rows_per_block - 1) / p->rows_per_block);
  if (p->grid_x < 1) p->grid_x = 1;
  p->lds_scores = (size_t)d * 32 * sizeof(float) + (size_t)kBW * kStageBytes;
  // top-k pass: slabs of >= 16 Ki rows, enough blocks to fill the chip
  long sl = (256L * 8 + nq - 1) / nq;
  long sl_max = (n + 16383) / 16384;
  if (sl > sl_max) sl = sl_max;
  if (sl < 1) sl = 1;
  p->rows_per_slab = ((n + sl - 1) / sl + 63) / 64 * 64;
  p->slabs = (int)((n + p->rows_per_slab - 1) / p->rows_per_slab);
  if (p->slabs < 1) p->slabs = 1;
  p->cap = topk_cap(k);
  p->s_bytes = (size_t)nq * (size_t)n * sizeof(float);
  p->part_bytes = (size_t)p->slabs * nq * k * sizeof(C32);
}

template <int D8>
static void launch_scores(const DenseMfmaPlan& p, const float* X, long n, const float* Q, int nq, float* S,
                          hipStream_t st) {
  static bool attr_done = false;  // 96-128 KiB of dynamic LDS needs the opt-in once per kernel
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)dense_mfma_scores_kernel<D8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              D8 * 8 * 32 * (int)sizeof(float) + kBW * kStageBytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((dense_mfma_scores_kernel<D8>), dim3(p.grid_x, p.q_tiles), dim3(256), p.lds_scores, st, X, n, Q,
                     nq, p.rows_per_block, S);
}

int dense_mfma_launch_scores(const DenseMfmaPlan& p, const float* X, long n, int d, const float* Q, int nq, float* S,
                             hipStream_t st) {
  switch (d) {
    case 384: launch_scores<48>(p, X, n, Q, nq, S, st); break;
    case 768: launch_scores<96>(p, X, n, Q, nq, S, st); break;
    case 1024: launch_scores<128>(p, X, n, Q, nq, S, st); break;
    default: return fail(AMDR_EINVAL, "dense (batched): unsupported dim %d", d);
  }
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int dense_mfma_launch_topk(const DenseMfmaPlan& p, const float* S, long n, int nq, int k, void* part,
                           float* fin_scores, int64_t* fin_ids, hipStream_t st) {
  const int waves = p.rows_per_slab <= 1024 ? 1 : kBW;
  size_t lds = (size_t)waves * p.cap * sizeof(C32) + waves * sizeof(int);
  if (waves == 1)
    hipLaunchKernelGGL(scores_slab_topk_kernel<1>, dim3(p.slabs, nq), dim3(64), lds, st, S, n, nq, k, p.cap,
                       p.rows_per_slab, (C32*)part, fin_scores, (long long*)fin_ids);
  else
    hipLaunchKernelGGL(scores_slab_topk_kernel<kBW>, dim3(p.slabs, nq), dim3(256), lds, st, S, n, nq, k, p.cap,
                       p.rows_per_slab, (C32*)part, fin_scores, (long long*)fin_ids);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

}  // namespace amdr
