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
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>

namespace amdr {


#define AMDR_MS_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AMDR_MS_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

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

// ---- split-fp16 form ("f16x3") -------------------------------------------------------------------------------
// The same tile on the fp16 matrix instructions (v_mfma_f32_32x32x16_f16: 16x the rate of the fp32-input form).
// fp16 alone (11 significant bits) would miss the 1e-4 bar, so every operand x (scaled by a power of two into
// [-1, 1], see below) is split EXACTLY into
//     x = hi + lo / 2048,   hi = fp16(x),   lo = fp16((x - hi) * 2048)
// (x - hi is exact in fp32; lo keeps its next 11 bits: 22 significant bits in all), and a product is taken as
//     a*b ~= a_hi*b_hi + (a_hi*b_lo + a_lo*b_hi) / 2048
// — three fp16 MFMAs instead of the fp32-input sequence, every fp16 x fp16 product exact in the fp32 accumulator;
// only the a_lo*b_lo term (2^-22 of the product) is dropped.  The cross terms run in their own accumulator and are
// folded in with one fma per score.  Measured on the UCC-en / Civil-Code-zh token stores against the fp64 oracle:
// max |score error| 2.0e-6 / 2.5e-6 on scores of ~20 (the fp32-input form: 3.7e-6 / 2.8e-6 — its 128-term fp32
// chains round more often), ranks identical (tests/test_kernels_gpu.py).  Document tokens are split ONCE at index
// creation into a [hi 128 x fp16 | lo 128 x fp16] image of the same 512 bytes per token as the fp32 row (same LDS
// tile, same swizzle); a query is split by its wave at kernel start.  Power-of-two scales (the store's: from its
// largest |component| at creation; a query's: from its own) keep hi / lo inside fp16's range for any finite input
// and are undone exactly on the per-token maxima.  AMDR_MAXSIM_F16X3=0 pins the fp32-input form.
//
// Shape: one 32x32x16 accumulator (16 registers) has the query token on the lane (l & 31) and 16 document tokens in
// the lane's registers: rows (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
//   A (32 doc tokens x 16):   lane (r = l & 31, h = l >> 5) holds row r, k = 16 s + 8 h .. + 7 = 16-B chunk 2 s + h
//   B (16 x 32 query tokens): the same of query-token row r.
// An MFMA holds its SIMD's vector issue for 8 cycles whatever its shape (MI355X_MICROARCH.md): 8 of 16 for a
// 16x16x32, 8 of 32 for a 32x32x16 — the first version of this form ran 48 16x16x32 MFMAs per tile and was bound by
// the issue port (PMC: ~100 vector + 50 scalar instructions per wave and tile beside them, matrix pipe 51 % busy;
// staggering the two waves of a SIMD by half a step gained 10 %); 24 of the wide shape leave 3x the issue slots:
// 2.40 -> 2.23 ms per 1 168 UCC-en queries (fp32-input form: 6.78 ms).  What bounds it now is POWER: under this kernel
// the chip holds 1.70 GHz (GRBM_GUI_ACTIVE; 2.11 GHz under the fp32-input dense kernel), the matrix pipe is busy 67 %
// of those cycles (77 % with DMA and barriers taken out in a timing-only build, which runs 2.10 ms); at the held
// clock the MFMAs alone need 1.63 ms.
typedef _Float16 ms8h __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kMsLoScale = 2048.f, kMsLoInv = 1.f / 2048.f;

__device__ __forceinline__ void ms_split(const float (&x)[8], float s, ms8h& hi, ms8h& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = x[j] * s;
    const _Float16 h = (_Float16)v;
    hi[j] = h;
    lo[j] = (_Float16)((v - (float)h) * kMsLoScale);
  }
}

// A query's fragments, split (lane (r, h): token row r, chunks 2 s + h); rows past q_len are zero; the wave's
// power-of-two scale comes back in `unscale` (1 / scale, exact).
__device__ __forceinline__ void ms_load_query_h(const float* __restrict__ Qq, int q_len, bool live, int r, int h,
                                                ms8h (&qh)[8], ms8h (&ql)[8], float& unscale) {
  float x[8][8];
  float m = 0.f;
  const bool out = !live || r >= q_len;
  const float* p = Qq + (size_t)(r < q_len ? r : q_len - 1) * kDim + 8 * h;
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    ms4f v0 = ms4f{0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (live) {
      v0 = *reinterpret_cast<const ms4f*>(p + 16 * st);
      v1 = *reinterpret_cast<const ms4f*>(p + 16 * st + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[st][j] = out ? 0.f : v0[j];
      x[st][4 + j] = out ? 0.f : v1[j];
      m = fmaxf(m, fmaxf(fabsf(x[st][j]), fabsf(x[st][4 + j])));
    }
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) m = fmaxf(m, __shfl_xor(m, sft));
  int e = 0;
  if (m > 0.f && m <= FLT_MAX) (void)frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
  const float sc = ldexpf(1.f, -e);
  unscale = ldexpf(1.f, e);
#pragma unroll
  for (int st = 0; st < 8; ++st) ms_split(x[st], sc, qh[st], ql[st]);
}

// One 32 x 32 tile: 24 MFMAs, then the lane's maximum over its 16 document tokens (rows >= remain are no tokens of
// the document: masked on the last tile of a document only — a real, wave-uniform branch: if-converted, the 32
// compare / select pairs ran on every tile).  Both kernels of this form call it: identical bits.
__device__ __forceinline__ void ms_tile_h(const ms8h (&ah)[8], const ms8h (&al)[8], const ms8h (&qh)[8],
                                          const ms8h (&ql)[8], int h, int remain, float& best) {
  f32x16 am, ac;
#pragma unroll
  for (int j = 0; j < 16; ++j) am[j] = ac[j] = 0.f;
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], qh[st], am, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], ql[st], ac, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st], qh[st], ac, 0, 0, 0);
  }
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(ac[j], kMsLoInv, am[j]);
  if (remain < 32) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (((j & 3) + 8 * (j >> 2) + 4 * h) >= remain) v[j] = -FLT_MAX;
  }
#pragma unroll
  for (int j = 0; j < 16; j += 2) best = fmaxf(best, fmaxf(v[j], v[j + 1]));
}

// Document score from the per-lane maxima: max over the two row halves, sum over the q_len query tokens (lanes
// 0..31 of h = 0), scales undone (powers of two: exact).
__device__ __forceinline__ float ms_finish_h(float best, int r, int h, int q_len, float unscale) {
  const float b = fmaxf(best, __uint_as_float(lane_xor<32>(__float_as_uint(best))));
  return ms_wave_sum((h == 0 && r < q_len) ? b * unscale : 0.f);
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

// fp32 token rows -> the [hi | lo] image (one thread per 8 components), scaled by the store's power of two
__global__ __launch_bounds__(256) void ms_split_store_kernel(const float* __restrict__ D, long n_tokens, float scale,
                                                             unsigned char* __restrict__ img,
                                                             unsigned char* __restrict__ img_hi /* [tok][128 x fp16] */) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (token, group of 8 components)
  if (i >= n_tokens * 16) return;
  const long tok = i >> 4;
  const int g = (int)(i & 15);
  float x[8];
  const ms4f v0 = *reinterpret_cast<const ms4f*>(D + tok * kDim + 8 * g);
  const ms4f v1 = *reinterpret_cast<const ms4f*>(D + tok * kDim + 8 * g + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = v0[j], x[4 + j] = v1[j];
  ms8h hi, lo;
  ms_split(x, scale, hi, lo);
  *reinterpret_cast<ms8h*>(img + tok * 512 + 16 * g) = hi;
  *reinterpret_cast<ms8h*>(img + tok * 512 + 256 + 16 * g) = lo;
  *reinterpret_cast<ms8h*>(img_hi + tok * 256 + 16 * g) = hi;
}

// largest token L2 norm of the store, as float bits (error bound of the first pass of the two-pass top-k)
__global__ __launch_bounds__(256) void ms_tokmax_kernel(const float* __restrict__ D, long n_tokens,
                                                        unsigned int* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wv = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (long)gridDim.x * 4;
  float m = 0.f;
  for (long t = wv; t < n_tokens; t += nw) {
    const float a = D[t * kDim + lane], b = D[t * kDim + 64 + lane];
    float ss = a * a + b * b;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) ss += __shfl_xor(ss, sft);
    m = fmaxf(m, sqrtf(ss));
  }
  if (lane == 0) atomicMax(out, __float_as_uint(m));
}

// largest |component| of the store as float bits (non-negative floats order like unsigned integers; NaN sorts
// above infinity, so a non-finite store is visible in the result)
__global__ __launch_bounds__(256) void ms_absmax_kernel(const float* __restrict__ D, long n, unsigned int* __restrict__ out) {
  unsigned int m = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    m = max(m, __float_as_uint(D[i]) & 0x7fffffffu);
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) m = max(m, (unsigned int)__shfl_xor((int)m, sft));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
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

// grid: (x = ceil(n_docs / 4), y = queries); one wave per document (1-7 queries: latency form), fp32-input form.
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

// The same, split-fp16 form: fragments straight from the [hi | lo] image (rows past the end of the document read on
// into the next document's tokens — the image is padded by one tile — and are masked by ms_tile_h).
__global__ __launch_bounds__(256) void maxsim_scores_h_kernel(const unsigned char* __restrict__ img,
                                                               const long long* __restrict__ doc_ptr, long n_docs,
                                                               const float* __restrict__ Q, int q_len,
                                                               float* __restrict__ scores /*[nq, n_docs]*/,
                                                               float unscale_d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long doc = (long)blockIdx.x * kMsWaves + wave;
  if (doc >= n_docs) return;
  const int qi = blockIdx.y;
  const int r32 = lane & 31, h = lane >> 5;
  ms8h qh[8], ql[8];
  float unscale;
  ms_load_query_h(Q + (size_t)qi * q_len * kDim, q_len, true, r32, h, qh, ql, unscale);
  unscale *= unscale_d;
  const long t_lo = doc_ptr[doc];
  const int len = (int)(doc_ptr[doc + 1] - t_lo);
  float best = -FLT_MAX;
  for (int tok0 = 0; tok0 < len; tok0 += 32) {
    const unsigned char* p = img + (size_t)(t_lo + tok0 + r32) * 512 + h * 16;
    ms8h ah[8], al[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      ah[st] = *reinterpret_cast<const ms8h*>(p + 32 * st);
      al[st] = *reinterpret_cast<const ms8h*>(p + 256 + 32 * st);
    }
    ms_tile_h(ah, al, qh, ql, h, len - tok0, best);
  }
  const float total = ms_finish_h(best, r32, h, q_len, unscale);
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

// ---- ring form of the blocked kernel (split-fp16 tiles) ---------------------------------------------------------
// A wave's 24 MFMAs of a split-fp16 tile take 768 cycles — less than one trip to L2 / the Infinity Cache — so the
// one-tile-ahead, register-staged pipeline of the kernel above cannot feed them.  Here the document tiles go through
// a RING of NBUF 16-KiB LDS stages filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write),
// NBUF - 1 tiles ahead of the one being multiplied:
//   step s:  s_waitcnt lgkmcnt(0) vmcnt(2 x tiles in flight behind tile s)   this wave's pieces of tile s have landed
//            s_barrier                                                       ... everybody's; tile s - 1 has been read
//            DMA of tile s + NBUF - 1 into the stage of tile s - 1
//            ds_reads of tile s; 24 MFMAs; per-lane maxima; (last tile of a document) the score
// One raw barrier per tile and no vmcnt(0) in the loop (a __syncthreads() would drain the DMAs in flight); the waits
// are the s_waitcnt BUILTIN, not inline asm: hipcc's own wait-count pass must see them, or it re-waits in front of
// the MFMAs.  A DMA lands lane-linear (stage base + lane * 16): the XOR swizzle that makes the ds_read_b128 fragment
// reads conflict-free is applied to the per-lane SOURCE address (dense_panel.hip does the same).  128 VGPRs and
// NBUF x 16 KiB of LDS: two blocks = four waves per SIMD per CU, so one block's barrier / DMA wait runs under the
// other's MFMAs.  Tried on the way (same-box A/B, scripts/ab_maxsim.py): 2 / 3 / 4 / 6 stages 2.55 / 2.31 / 2.24 /
// 2.44 ms per 1 168 UCC-en queries; 8 / 16 / 32 / 64 documents per block 2.27 / 2.23 / 2.24 / 2.37; a second fragment
// register set filled one tile ahead (254 VGPRs) +- 0.
template <int NBUF>
__global__ __launch_bounds__(kMsQ * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void maxsim_scores_ring_kernel(const unsigned char* __restrict__ img,
                                                                        const long long* __restrict__ doc_ptr,
                                                                        long n_docs, int docs_per_block,
                                                                        const float* __restrict__ Q, int nq, int q_len,
                                                                        float* __restrict__ scores /*[nq, n_docs]*/,
                                                                        float unscale_d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // [NBUF][32 * 512]
  constexpr int kStage = 32 * 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int qi = blockIdx.x * kMsQ + wave;
  const bool live = qi < nq;
  const long d0 = (long)blockIdx.y * docs_per_block;
  long d1 = d0 + docs_per_block;
  if (d1 > n_docs) d1 = n_docs;

  ms8h qh[8], ql[8];
  float unscale;
  ms_load_query_h(Q + (size_t)(live ? qi : 0) * q_len * kDim, q_len, live, r32, h, qh, ql, unscale);
  unscale *= unscale_d;

  // DMA role: piece u (0, 1) of this wave covers stage bytes [(2 wave + u) * 1024, + 1024) = tile rows
  // 2 (2 wave + u) and + 1; lane l: row + (l >> 5), PHYSICAL slot l & 31, which holds logical slot ^ (row & 15).
  // poff: the piece's per-lane byte offset inside a tile; rows past the end of a document read on into the next
  // document's tokens (the image is padded by one tile at its end) and are masked after the MFMAs.
  long poff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int prow = 2 * (2 * wave + u) + (lane >> 5);
    poff[u] = (long)prow * 512 + (((lane & 31) ^ (prow & 15)) << 4);
  }
  // fragment read addresses: row r32, chunk 2 s + h (the lo part sits 256 B behind the hi part: slots c and 16 + c
  // differ in bit 4, which the XOR with row & 15 leaves alone)
  int foff[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) foff[st] = ms_tile_off(r32, 2 * st + h);

  struct Cur {  // a position in the block's tile sequence (wave-uniform)
    long doc, t_lo;
    int len, tok0;
  };
  auto advance = [&](Cur& c) {
    c.tok0 += 32;
    if (c.tok0 >= c.len) {
      c.doc += 1;
      c.tok0 = 0;
      if (c.doc < d1) {
        c.t_lo = doc_ptr[c.doc];
        c.len = (int)(doc_ptr[c.doc + 1] - c.t_lo);
      }
    }
  };
  auto issue = [&](const Cur& c, int stage) {
    const unsigned char* src = img + (size_t)(c.t_lo + c.tok0) * 512;  // wave-uniform
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds(AMDR_MS_GPTR(src + poff[u]),
                                       AMDR_MS_LPTR(ring + stage * kStage + (2 * wave + u) * 1024), 16, 0, 0);
  };
  Cur prod, cur;
  prod.doc = d0;
  prod.t_lo = doc_ptr[d0];
  prod.len = (int)(doc_ptr[d0 + 1] - prod.t_lo);
  prod.tok0 = 0;
  cur = prod;
  int issued = 0, done = 0;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) {
    if (prod.doc < d1) {
      issue(prod, issued % NBUF);
      ++issued;
      advance(prod);
    }
  }
  float best = -FLT_MAX;
  while (cur.doc < d1) {
    // simm16 on gfx9: vmcnt [3:0] (+ [15:14]), expcnt [6:4] (7 = none), lgkmcnt [11:8].  lgkmcnt(0) on every path as
    // ONE unconditional instruction (inside the branches below the pass still re-waited in front of the MFMAs);
    // this wave's pieces of tile `done` have landed once at most 2 x (tiles issued after it) loads are outstanding.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int behind = issued - done - 1;
    if (behind >= 3) {
      __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
    } else if (behind == 2) {
      __builtin_amdgcn_s_waitcnt(0x0F74);
    } else if (behind == 1) {
      __builtin_amdgcn_s_waitcnt(0x0F72);
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prod.doc < d1) {
      issue(prod, issued % NBUF);
      ++issued;
      advance(prod);
    }
    const unsigned char* tile = ring + (done % NBUF) * kStage;
    ms8h ah[8], al[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const unsigned char* fp = tile + foff[st];
      ah[st] = *reinterpret_cast<const ms8h*>(fp);
      al[st] = *reinterpret_cast<const ms8h*>(fp + 256);
    }
    const int remain = cur.len - cur.tok0;
    ms_tile_h(ah, al, qh, ql, h, remain, best);
    if (remain <= 32) {  // last tile of this document
      const float total = ms_finish_h(best, r32, h, q_len, unscale);
      if (live && lane == 0) scores[(size_t)qi * n_docs + cur.doc] = total;
      best = -FLT_MAX;
    }
    ++done;
    advance(cur);
  }
}

// ---- two-pass top-k: a cheap first pass picks the documents worth the full arithmetic ---------------------------
// `search` needs the k best documents, not every score.  Pass 1 scores every (query, document) with the hi parts only
// (ONE fp16 MFMA per block instead of three, a 256-byte-per-token image instead of 512: a third of the matrix cycles and
// half the bytes).  |a_hi . b_hi - a . b| <= (2^-10 + 2^-22) |a| |b| (each fp16 rounding is 2^-11 relative, per
// component), so a document's first-pass score is within
//     eps_q = 1.5 * 2^-10 * (sum_i |q_i|) * max_token |d|  (+ the subnormal term)
// of its full-form score (the 1.5 covers the fp32 accumulation and the full form's own 2e-6).  If T is the k-th best
// first-pass score, every document that can be among the k best full-form scores — ties at the cut included — has a
// first-pass score >= T - 2 eps_q (at most k - 1 documents score above the k-th best s_k, hence T <= s_k + eps, and a
// top-k document has a >= s_k - eps >= T - 2 eps).  Pass 2 re-scores exactly those documents with the full form
// (the tile function of the one-pass kernels: the same bits) and the final top-k runs on the re-scored values:
// identical ids and scores by construction, and by test against the one-pass form.  On the UCC-en / Civil-Code-zh
// stores 11-13 of 591 / 1 260 documents per query pass the cut at k = 10 (about 90 at k = 80).  A query with more than
// `cap` candidates (mass near-ties) re-scores every document instead (maxsim_overflow_kernel).
// Pass 1: the ring kernel above with 64-token tiles of the hi-only image (16-KiB stages again, half the barriers per
// document), 16 MFMAs and 16 ds_read_b128 per tile, no fma in the epilogue.
__device__ __forceinline__ int ms_hi_off(int row, int slot) { return row * 256 + ((slot ^ (row & 15)) << 4); }

template <int NBUF>
__global__ __launch_bounds__(kMsQ * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void maxsim_hi_ring_kernel(
    const unsigned char* __restrict__ img_hi, const long long* __restrict__ doc_ptr, long n_docs, int docs_per_block,
    const float* __restrict__ Q, int nq, int q_len, float* __restrict__ approx /*[nq, n_docs]*/, float unscale_d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // [NBUF][64 * 256]
  constexpr int kStage = 64 * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int qi = blockIdx.x * kMsQ + wave;
  const bool live = qi < nq;
  const long d0 = (long)blockIdx.y * docs_per_block;
  long d1 = d0 + docs_per_block;
  if (d1 > n_docs) d1 = n_docs;

  ms8h qh[8], ql_unused[8];
  float unscale;
  ms_load_query_h(Q + (size_t)(live ? qi : 0) * q_len * kDim, q_len, live, r32, h, qh, ql_unused, unscale);
  unscale *= unscale_d;

  // DMA role: piece u (0, 1) of this wave covers stage bytes [(2 wave + u) * 1024, + 1024) = tile rows 4 (2 wave + u) .. + 3
  // (256 B each); lane l: row + (l >> 4), PHYSICAL slot l & 15, which holds logical slot ^ (row & 15)
  long poff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int prow = 4 * (2 * wave + u) + (lane >> 4);
    poff[u] = (long)prow * 256 + (((lane & 15) ^ (prow & 15)) << 4);
  }
  int foff[8];  // fragment reads: row r32 (+ 32 for the second row block: 32 * 256 B further, same row & 15), chunk 2 s + h
#pragma unroll
  for (int st = 0; st < 8; ++st) foff[st] = ms_hi_off(r32, 2 * st + h);

  struct Cur {
    long doc, t_lo;
    int len, tok0;
  };
  auto advance = [&](Cur& c) {
    c.tok0 += 64;
    if (c.tok0 >= c.len) {
      c.doc += 1;
      c.tok0 = 0;
      if (c.doc < d1) {
        c.t_lo = doc_ptr[c.doc];
        c.len = (int)(doc_ptr[c.doc + 1] - c.t_lo);
      }
    }
  };
  auto issue = [&](const Cur& c, int stage) {
    const unsigned char* src = img_hi + (size_t)(c.t_lo + c.tok0) * 256;  // wave-uniform
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds(AMDR_MS_GPTR(src + poff[u]),
                                       AMDR_MS_LPTR(ring + stage * kStage + (2 * wave + u) * 1024), 16, 0, 0);
  };
  Cur prod, cur;
  prod.doc = d0;
  prod.t_lo = doc_ptr[d0];
  prod.len = (int)(doc_ptr[d0 + 1] - prod.t_lo);
  prod.tok0 = 0;
  cur = prod;
  int issued = 0, done = 0;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) {
    if (prod.doc < d1) {
      issue(prod, issued % NBUF);
      ++issued;
      advance(prod);
    }
  }
  float best = -FLT_MAX;
  while (cur.doc < d1) {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); then this wave's pieces of tile `done` (see maxsim_scores_ring_kernel)
    const int behind = issued - done - 1;
    if (behind >= 3) {
      __builtin_amdgcn_s_waitcnt(0x0F76);
    } else if (behind == 2) {
      __builtin_amdgcn_s_waitcnt(0x0F74);
    } else if (behind == 1) {
      __builtin_amdgcn_s_waitcnt(0x0F72);
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prod.doc < d1) {
      issue(prod, issued % NBUF);
      ++issued;
      advance(prod);
    }
    const unsigned char* tile = ring + (done % NBUF) * kStage;
    const int remain = cur.len - cur.tok0;
    ms8h a0[8], a1[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) a0[st] = *reinterpret_cast<const ms8h*>(tile + foff[st]);
    f32x16 c0, c1;
#pragma unroll
    for (int j = 0; j < 16; ++j) c0[j] = c1[j] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[st], qh[st], c0, 0, 0, 0);
    if (remain > 32) {  // wave-uniform: the second 32-token row block holds tokens of this document
#pragma unroll
      for (int st = 0; st < 8; ++st) a1[st] = *reinterpret_cast<const ms8h*>(tile + 32 * 256 + foff[st]);
#pragma unroll
      for (int st = 0; st < 8; ++st) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[st], qh[st], c1, 0, 0, 0);
    }
    if (remain < 64) {  // last tile of a document: rows >= remain are no tokens of it
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
        if (row >= remain) c0[j] = -FLT_MAX;
        if (32 + row >= remain) c1[j] = -FLT_MAX;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) best = fmaxf(best, fmaxf(c0[j], c1[j]));
    if (remain <= 64) {
      const float total = ms_finish_h(best, r32, h, q_len, unscale);
      if (live && lane == 0) approx[(size_t)qi * n_docs + cur.doc] = total;
      best = -FLT_MAX;
    }
    ++done;
    advance(cur);
  }
}

// Pass 1, two queries per wave.  PMC / arithmetic on the kernel above: a wave reads the whole 16-KiB tile from LDS for 16
// MFMAs of 32 cycles — 16 waves per CU x 16 KiB per 2 048 pipe cycles = 125 B per clock, the LDS's whole bandwidth: it
// ran at half its matrix floor (1.0 ms against 0.52).  Here a wave keeps the hi fragments of TWO queries (64 VGPRs) and
// feeds both from one read of the tile: half the LDS bytes per MFMA.  A block = 4 waves = 8 queries (the tile is shared
// by as many queries as before), 3 stages = 48 KiB, three blocks per CU.
constexpr int kMsQ2 = 4;  // waves per block; 2 queries each

__device__ __forceinline__ void ms_load_query_img_hi(const unsigned char* __restrict__ img_q, int qi, int r32, int h,
                                                     ms8h (&qh)[8]) {
  const unsigned char* p = img_q + ((size_t)qi * 32 + r32) * 512 + 16 * h;
#pragma unroll
  for (int st = 0; st < 8; ++st) qh[st] = *reinterpret_cast<const ms8h*>(p + 32 * st);
}

__device__ __forceinline__ float ms_max3(float a, float b, float c) {  // max(a, b, c) in one instruction (no NaNs reach it)
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int NBUF>
__global__ __launch_bounds__(kMsQ2 * 64) __attribute__((amdgpu_waves_per_eu(3, NBUF == 2 ? 5 : 3))) void maxsim_hi2_ring_kernel(
    const unsigned char* __restrict__ img_hi, const long long* __restrict__ doc_ptr, long n_docs, int docs_per_block,
    const float* __restrict__ Q, int nq, int q_len, float* __restrict__ approx /*[nq, n_docs]*/, float unscale_d,
    const unsigned char* __restrict__ img_q /* nullable: the queries' split images (maxsim_split_queries_kernel) */,
    const float* __restrict__ unscale_q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // [NBUF][64 * 256]
  constexpr int kStage = 64 * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int qa = (blockIdx.x * kMsQ2 + wave) * 2, qb = qa + 1;
  const bool live_a = qa < nq, live_b = qb < nq;
  const long d0 = (long)blockIdx.y * docs_per_block;
  long d1 = d0 + docs_per_block;
  if (d1 > n_docs) d1 = n_docs;

  ms8h qha[8], qhb[8];
  float unscale_a, unscale_b;
  if (img_q) {
    // the fragments as maxsim_split_queries_kernel left them (a query is scored by n_docs / docs_per_block blocks:
    // splitting it in each of them was 10-20 % of this kernel's vector instructions); a dead wave takes query 0's
    ms_load_query_img_hi(img_q, live_a ? qa : 0, r32, h, qha);
    ms_load_query_img_hi(img_q, live_b ? qb : 0, r32, h, qhb);
    unscale_a = unscale_q[live_a ? qa : 0];
    unscale_b = unscale_q[live_b ? qb : 0];
  } else {
    ms8h lo_unused[8];
    ms_load_query_h(Q + (size_t)(live_a ? qa : 0) * q_len * kDim, q_len, live_a, r32, h, qha, lo_unused, unscale_a);
    ms_load_query_h(Q + (size_t)(live_b ? qb : 0) * q_len * kDim, q_len, live_b, r32, h, qhb, lo_unused, unscale_b);
  }
  unscale_a *= unscale_d;
  unscale_b *= unscale_d;

  // DMA role: pieces 4 wave .. 4 wave + 3 of the tile's 16 (1 KiB = 4 rows of 256 B each); lane l: row + (l >> 4),
  // PHYSICAL slot l & 15, which holds logical slot ^ (row & 15)
  long poff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int prow = 4 * (4 * wave + u) + (lane >> 4);
    poff[u] = (long)prow * 256 + (((lane & 15) ^ (prow & 15)) << 4);
  }
  int foff[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) foff[st] = ms_hi_off(r32, 2 * st + h);

  struct Cur {
    long doc, t_lo;
    int len, tok0;
  };
  auto advance = [&](Cur& c) {
    c.tok0 += 64;
    if (c.tok0 >= c.len) {
      c.doc += 1;
      c.tok0 = 0;
      if (c.doc < d1) {
        c.t_lo = doc_ptr[c.doc];
        c.len = (int)(doc_ptr[c.doc + 1] - c.t_lo);
      }
    }
  };
  auto issue = [&](const Cur& c, int stage) {
    const unsigned char* src = img_hi + (size_t)(c.t_lo + c.tok0) * 256;  // wave-uniform
#pragma unroll
    for (int u = 0; u < 4; ++u)
      __builtin_amdgcn_global_load_lds(AMDR_MS_GPTR(src + poff[u]),
                                       AMDR_MS_LPTR(ring + stage * kStage + (4 * wave + u) * 1024), 16, 0, 0);
  };
  Cur prod, cur;
  prod.doc = d0;
  prod.t_lo = doc_ptr[d0];
  prod.len = (int)(doc_ptr[d0 + 1] - prod.t_lo);
  prod.tok0 = 0;
  cur = prod;
  int issued = 0, done = 0;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) {
    if (prod.doc < d1) {
      issue(prod, issued % NBUF);
      ++issued;
      advance(prod);
    }
  }
  float best_a = -FLT_MAX, best_b = -FLT_MAX;
  while (cur.doc < d1) {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); then this wave's 4 pieces of tile `done` (4 loads per tile in flight behind it)
    const int behind = issued - done - 1;
    if (behind >= 2) {
      __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8)
    } else if (behind == 1) {
      __builtin_amdgcn_s_waitcnt(0x0F74);  // vmcnt(4)
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prod.doc < d1) {
      issue(prod, issued % NBUF);
      ++issued;
      advance(prod);
    }
    const unsigned char* tile = ring + (done % NBUF) * kStage;
    const int remain = cur.len - cur.tok0;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      if (blk == 1 && remain <= 32) break;  // wave-uniform: the second 32-token row block holds no token of this document
      ms8h a[8];
#pragma unroll
      for (int st = 0; st < 8; ++st) a[st] = *reinterpret_cast<const ms8h*>(tile + blk * (32 * 256) + foff[st]);
      f32x16 ca, cb;
#pragma unroll
      for (int j = 0; j < 16; ++j) ca[j] = cb[j] = 0.f;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st], qha[st], ca, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st], qhb[st], cb, 0, 0, 0);
      }
      if (remain < 32 * (blk + 1)) {  // last row block of a document: rows >= remain are no tokens of it
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (32 * blk + (j & 3) + 8 * (j >> 2) + 4 * h >= remain) ca[j] = cb[j] = -FLT_MAX;
      }
      // 16 values -> 1 per query and row block as EIGHT v_max3_f32 (hipcc fused only a quarter of the fmaxf pairs: 56
      // v_max per 32 MFMAs; the pass issued ~5 other vector instructions per MFMA with its matrix pipe busy half the
      // time: 975 -> 891 us per 1 168 UCC-en queries).  Tried after it and dropped: FOUR queries per wave (64 MFMAs per
      // tile read and barrier, 16 queries per tile, two waves per SIMD): 1.08 against 1.09 ms per hybrid step.  Also
      // tried and dropped: the fragment reads of the NEXT 32-token row block issued before the current block's 32 MFMAs
      // (the compiler's order here is two reads, s_waitcnt lgkmcnt(0), four MFMAs, eight times per tile) with a second
      // fragment set — 64 more VGPRs, two waves per SIMD, four ring stages: 1.136 against 1.101 ms per hybrid step, slower
      // at every block size (scripts/sweep_maxsim_docs.sh).  Neither the reads per MFMA nor their latency is what keeps
      // the matrix pipe at half duty.
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        best_a = ms_max3(best_a, ca[j], ca[j + 1]);
        best_b = ms_max3(best_b, cb[j], cb[j + 1]);
      }
    }
    if (remain <= 64) {
      const float ta = ms_finish_h(best_a, r32, h, q_len, unscale_a);
      const float tb = ms_finish_h(best_b, r32, h, q_len, unscale_b);
      if (lane == 0) {
        if (live_a) approx[(size_t)qa * n_docs + cur.doc] = ta;
        if (live_b) approx[(size_t)qb * n_docs + cur.doc] = tb;
      }
      best_a = best_b = -FLT_MAX;
    }
    ++done;
    advance(cur);
  }
}

// One wave splits one query into the [hi | lo] image of the re-scoring pass (512 B per token row, 32 rows, rows past q_len
// zero) and stores its power-of-two unscale — the scale rule and ms_split of ms_load_query_h: identical fragments.  The
// re-scoring pass takes a query's fragments for every (document, query) item it serves, 15 k times per UCC-en batch:
// splitting them in the scoring wave each time cost ~500 vector instructions per item and wave.
// `norm_sum` (nullable): the sum of the token rows' Euclidean norms, which the candidate margin of the two-pass top-k is
// built on (maxsim_select_kernel) — the rows are in this wave's registers anyway.
__device__ __forceinline__ void ms_split_query_wave(const float* __restrict__ Qq, int q_len, int lane,
                                                    unsigned char* __restrict__ img, float* __restrict__ unscale,
                                                    float* __restrict__ norm_sum = nullptr) {
  // one pass over the query: lane holds (row, group of 8 components) g = lane + 64 it — 16 lanes per row, 4 rows per step
  float x[8][8];
  float m = 0.f, nsum = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int g = lane + 64 * it, row = g >> 4, grp = g & 15;
    float ss = 0.f;
    if (row < q_len) {
      const ms4f v0 = *reinterpret_cast<const ms4f*>(Qq + (size_t)row * kDim + 8 * grp);
      const ms4f v1 = *reinterpret_cast<const ms4f*>(Qq + (size_t)row * kDim + 8 * grp + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[it][j] = v0[j], x[it][4 + j] = v1[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[it][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      m = fmaxf(m, fabsf(x[it][j]));
      ss += x[it][j] * x[it][j];
    }
#pragma unroll
    for (int sft = 1; sft < 16; sft <<= 1) ss += __shfl_xor(ss, sft);  // the row's 16 lanes
    nsum += sqrtf(ss);  // (every lane of the row holds it; counted once below)
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) m = fmaxf(m, __shfl_xor(m, sft));
  nsum += __shfl_xor(nsum, 16);  // the four rows of a step sit in the four 16-lane groups
  nsum += __shfl_xor(nsum, 32);
  int e = 0;
  if (m > 0.f && m <= FLT_MAX) (void)frexpf(m, &e);
  const float sc = ldexpf(1.f, -e);
  if (lane == 0) {
    *unscale = ldexpf(1.f, e);
    if (norm_sum) *norm_sum = nsum;
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int g = lane + 64 * it, row = g >> 4, grp = g & 15;
    ms8h hi, lo;
    ms_split(x[it], sc, hi, lo);
    unsigned char* dst = img + (size_t)row * 512 + 16 * grp;
    *reinterpret_cast<ms8h*>(dst) = hi;
    *reinterpret_cast<ms8h*>(dst + 256) = lo;
  }
}

// Ahead of pass 1 (round 4), one wave per query: both passes take their query fragments from these images.
__global__ __launch_bounds__(256) void maxsim_split_queries_kernel(const float* __restrict__ Q, int nq, int q_len,
                                                                   unsigned char* __restrict__ img_q,
                                                                   float* __restrict__ unscale_out,
                                                                   float* __restrict__ norm_sum) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  ms_split_query_wave(Q + (size_t)q * q_len * kDim, q_len, threadIdx.x & 63, img_q + (size_t)q * 32 * 512,
                      unscale_out + q, norm_sum + q);
}

// Between the passes, one wave per query: T = the k-th best first-pass score, eps from the query's token norms, the
// list of documents with a first-pass score >= T - 2 eps (ascending ids, at most cap; more -> overflow), and the
// re-scored row initialised to "not a candidate".
__global__ __launch_bounds__(64) void maxsim_select_kernel(const float* __restrict__ approx, long n_docs,
                                                           const float* __restrict__ Q, int q_len, int k, int cap_sel,
                                                           float d_norm_max, float unscale_d, int cap,
                                                           float* __restrict__ exact /*[nq, n_docs]*/,
                                                           int* __restrict__ cand /*[nq, cap]*/, int* __restrict__ cnt,
                                                           int* __restrict__ overflow, int* __restrict__ dcnt,
                                                           int* __restrict__ dlist /*[n_docs][nq]*/, int nq,
                                                           const float* __restrict__ norm_sum /* nullable: with */,
                                                           const float* __restrict__ unscale_in /* the split images */,
                                                           int init_exact) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C32* buf = reinterpret_cast<C32*>(smem);
  const int lane = threadIdx.x & 63, q = blockIdx.x;
  const float* row = approx + (size_t)q * n_docs;
  WaveTopK<C32> tk;
  tk.init(buf, cap_sel, k);
  // T: short rows in registers (the slab top-k's selector: 47 -> ~20 us per 1 168 UCC-en queries together with the two
  // changes below), otherwise — and on mass ties at the cut — the staged selector
  int got = -1;
  if (k <= 64 && n_docs <= kSelectRowsMax && cap_sel >= 128) {
    if (n_docs <= 640)
      got = select_row<10>(row, 0, n_docs, k, lane, tk.buf);
    else if (n_docs <= 1280)
      got = select_row<20>(row, 0, n_docs, k, lane, tk.buf);
    else
      got = select_row<32>(row, 0, n_docs, k, lane, tk.buf);
  }
  if (got >= 0) {
    tk.cnt = got;
  } else {
    for (long base = 0; base < n_docs; base += 64) {
      const long d = base + lane;
      const bool v = d < n_docs;
      tk.push_lanes(v ? C32::make(row[d], (u32)d) : C32::pad(), v, lane);
    }
    tk.finalize(lane);
  }
  wave_lds_fence();
  const float T = tk.cnt >= k ? tk.buf[k - 1].score() : -FLT_MAX;  // fewer than k documents: every one is a candidate
  wave_lds_fence();
  // eps: token norms and the query's power-of-two scale, as the scoring kernels take it
  float nsum = 0.f, unscale_q;
  if (norm_sum) {  // left by maxsim_split_queries_kernel
    nsum = norm_sum[q];
    unscale_q = unscale_in[q];
  } else {
    const float* Qq = Q + (size_t)q * q_len * kDim;
    float amax = 0.f;
    for (int i = 0; i < q_len; ++i) {
      const float a = Qq[i * kDim + lane], b = Qq[i * kDim + 64 + lane];
      float ss = a * a + b * b;
      amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) ss += __shfl_xor(ss, sft);
      nsum += sqrtf(ss);
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) amax = fmaxf(amax, __shfl_xor(amax, sft));
    int e = 0;
    if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e);
    unscale_q = ldexpf(1.f, e);
  }
  const float eps = 1.5f * 9.765625e-4f * nsum * d_norm_max * 1.0001f +
                    (float)q_len * 256.f * 2.98023224e-8f * unscale_q * unscale_d;  // + operands in fp16's subnormal range
  const float thr = (T == -FLT_MAX) ? -FLT_MAX : T - 2.f * eps;
  int n = 0;
  for (long base = 0; base < n_docs; base += 64) {
    const long d = base + lane;
    const bool v = d < n_docs;
    const float a = v ? row[d] : 0.f;
    const bool pass = v && (a >= thr || thr == -FLT_MAX);
    if (v && init_exact) exact[(size_t)q * n_docs + d] = -FLT_MAX;  // (only rowscores_topk_kernel reads whole rows)
    const unsigned long long m = __ballot(pass);
    const int at = n + __popcll(lane ? (m & (~0ull >> (64 - lane))) : 0ull);
    if (pass && at < cap) cand[(size_t)q * cap + at] = (int)d;
    n += __popcll(m);
  }
  if (lane == 0) {
    overflow[q] = n > cap ? 1 : 0;
    cnt[q] = n > cap ? 0 : n;
  }
  if (dcnt != nullptr && n <= cap) {
    // round 4: the re-scoring pass walks the pairs BY DOCUMENT — the query joins the list of each of its candidates
    // (row d of dlist, one slot per query at most: no offsets to compute, no second pass to fill them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own list, just written
    for (int j = lane; j < n; j += 64) {
      const int d = cand[(size_t)q * cap + j];
      dlist[(size_t)d * nq + atomicAdd(dcnt + d, 1)] = q;  // the order inside a document's list does not matter
    }
  }
}

// Full-form score of one (query, document) by ONE wave, the document's tiles staged through a wave-private 16-KiB LDS
// stage by LDS-DMA (16 pieces of 1 KiB per tile: whole 512-byte token rows per request).  The first version
// fetched the MFMA fragments straight from global memory, as maxsim_scores_h_kernel does for a single query: 16-byte
// pieces of 32 different rows per load instruction — 15 k candidate pairs per launch then moved ~8x their bytes through
// the L1s and pass 2 took as long as pass 1 (0.98 ms).  Same tile function, same operands: the same bits.
__device__ __forceinline__ float ms_exact_doc_lds(const unsigned char* __restrict__ img,
                                                  const long long* __restrict__ doc_ptr, long doc, const ms8h (&qh)[8],
                                                  const ms8h (&ql)[8], int lane, int q_len, float unscale,
                                                  unsigned char* stage /* this wave's 16 KiB */) {
  const int r32 = lane & 31, h = lane >> 5;
  const long t_lo = doc_ptr[doc];
  const int len = (int)(doc_ptr[doc + 1] - t_lo);
  long poff[16];  // piece u: tile rows 2 u and 2 u + 1; lane: row + (l >> 5), PHYSICAL slot l & 31 <- logical slot ^ (row & 15)
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int prow = 2 * u + (lane >> 5);
    poff[u] = (long)prow * 512 + (((lane & 31) ^ (prow & 15)) << 4);
  }
  int foff[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) foff[st] = ms_tile_off(r32, 2 * st + h);
  float best = -FLT_MAX;
  for (int tok0 = 0; tok0 < len; tok0 += 32) {
    const unsigned char* src = img + (size_t)(t_lo + tok0) * 512;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // the fragment reads of the previous tile are done: the stage may be refilled
#pragma unroll
    for (int u = 0; u < 16; ++u)
      __builtin_amdgcn_global_load_lds(AMDR_MS_GPTR(src + poff[u]), AMDR_MS_LPTR(stage + u * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the tile has landed (the other waves of the CU cover the wait)
    asm volatile("" ::: "memory");
    ms8h ah[8], al[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const unsigned char* fp = stage + foff[st];
      ah[st] = *reinterpret_cast<const ms8h*>(fp);
      al[st] = *reinterpret_cast<const ms8h*>(fp + 256);
    }
    ms_tile_h(ah, al, qh, ql, h, len - tok0, best);
  }
  return ms_finish_h(best, r32, h, q_len, unscale);
}

// exclusive prefix sum of the candidate counts: off[q] = first item of query q in the flat work list, off[nq] = items
__global__ __launch_bounds__(256) void maxsim_offsets_kernel(const int* __restrict__ cnt, int nq, int* __restrict__ off) {
  __shared__ int part[256];
  int carry = 0;
  for (int base = 0; base < nq; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nq ? cnt[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int sft = 1; sft < 256; sft <<= 1) {
      const int o = threadIdx.x >= sft ? part[threadIdx.x - sft] : 0;
      __syncthreads();
      part[threadIdx.x] += o;
      __syncthreads();
    }
    if (i < nq) off[i] = carry + part[threadIdx.x] - v;
    carry += part[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) off[nq] = carry;
}

// Pass 2: the candidate (query, document) pairs as ONE flat work list cut into equal contiguous shares, one per wave of
// a grid that just fills the chip (a grid of (slot, query) blocks was mostly empty blocks queueing for LDS: 0.76 ms for
// 15 k pairs).  A share's items mostly belong to one query: its fragments are loaded once per query change.
__global__ __launch_bounds__(256) void maxsim_rescore_kernel(const unsigned char* __restrict__ img,
                                                             const long long* __restrict__ doc_ptr, long n_docs,
                                                             const float* __restrict__ Q, int nq, int q_len,
                                                             float unscale_d, const int* __restrict__ cand,
                                                             const int* __restrict__ off, int cap,
                                                             float* __restrict__ exact) {
  extern __shared__ __attribute__((aligned(16))) unsigned char stages[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int total = off[nq];
  const int G = gridDim.x * kMsWaves, g = blockIdx.x * kMsWaves + wave;
  const int share = (total + G - 1) / G;
  const int lo = g * share;
  int hi = lo + share;
  if (hi > total) hi = total;
  if (lo >= hi) return;  // whole wave
  int q = 0;
  {  // the query of item lo: the last q with off[q] <= lo
    int a = 0, b = nq;
    while (b - a > 1) {
      const int m = (a + b) >> 1;
      if (off[m] <= lo) a = m; else b = m;
    }
    q = a;
  }
  ms8h qh[8], ql[8];
  float unscale = 1.f;
  int q_loaded = -1;
  for (int item = lo; item < hi; ++item) {
    while (item >= off[q + 1]) ++q;  // queries without candidates are skipped
    if (q != q_loaded) {
      ms_load_query_h(Q + (size_t)q * q_len * kDim, q_len, true, lane & 31, lane >> 5, qh, ql, unscale);
      unscale *= unscale_d;
      q_loaded = q;
    }
    const long doc = cand[(size_t)q * cap + (item - off[q])];
    const float total_s = ms_exact_doc_lds(img, doc_ptr, doc, qh, ql, lane, q_len, unscale, stages + wave * 16384);
    if (lane == 0) exact[(size_t)q * n_docs + doc] = total_s;
  }
}

__device__ __forceinline__ void ms_load_query_img(const unsigned char* __restrict__ img_q, int qi, int r32, int h,
                                                  ms8h (&qh)[8], ms8h (&ql)[8]) {
  const unsigned char* p = img_q + ((size_t)qi * 32 + r32) * 512 + 16 * h;
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    qh[st] = *reinterpret_cast<const ms8h*>(p + 32 * st);
    ql[st] = *reinterpret_cast<const ms8h*>(p + 256 + 32 * st);
  }
}

// ---- Pass 2, round 4: the candidate pairs grouped BY DOCUMENT ------------------------------------------------------
// One wave per pair shared nothing: 15 k pairs x a document image of ~80 KB = 0.93 GB through the fabric for 68 MB of
// token store (PMC, profiles/r03_pmc.md), 145 us.  A document is a candidate of ~25 queries on the serving corpora, so
// the pairs are inverted to per-document query lists and a block takes (document, 8 of its queries): the document's
// tiles go ONCE through the block's LDS ring (the one-pass ring kernel's, same fragments, same tile function: the same
// bits) and feed 8 queries' MFMAs.
//   maxsim_select_kernel        also appends the query to the list of each of its candidate documents (dlist, dcnt)
//   maxsim_items_kernel         the item table, longest documents first (the first version of this round also
//                               prefix-summed per-document offsets and filled a packed pair list with a third kernel:
//                               fixed-stride lists written by the select kernel itself took both away)
//   maxsim_rescore_ring_kernel  persistent blocks of 8 waves walk the items
struct MsItem {  // one unit of the re-scoring pass: a document (its token range) and up to 8 of its queries
  int doc, p0, cnt, len;
  long long t_lo, pad;
};
__global__ __launch_bounds__(256) void maxsim_items_kernel(const int* __restrict__ dcnt, long n_docs,
                                                           const long long* __restrict__ doc_ptr,
                                                           int* __restrict__ n_items, MsItem* __restrict__ items) {
  __shared__ int bucket[16], bpos[16];
  if (threadIdx.x < 16) bucket[threadIdx.x] = 0;
  __syncthreads();
  // items per cost class: a document's items (8 of its queries each) cost its tiles
  for (long i = threadIdx.x; i < n_docs; i += 256) {
    const int it = (dcnt[i] + kMsQ - 1) / kMsQ;
    if (it) {
      const int tiles = (int)((doc_ptr[i + 1] - doc_ptr[i] + 31) >> 5);
      atomicAdd(&bucket[15 - (tiles < 15 ? tiles : 15)], it);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int at = 0;
    for (int c = 0; c < 16; ++c) {
      bpos[c] = at;
      at += bucket[c];
    }
    *n_items = at;
  }
  __syncthreads();
  // The item table, LONGEST DOCUMENTS FIRST (a counting sort over the tile count): the re-scoring blocks take items
  // b, b + grid, b + 2 grid, ... — dealt from a descending order every block's share costs about the same.  In document
  // order a block's 4-5 items ranged from 1 to 7 tiles each and the waves were alive for 65 % of the launch (SQ_WAVE_CYCLES).
  // One 32-byte descriptor per item (p0 = its first slot in the document's query list): a block reads it instead of
  // searching offsets.
  for (long i = threadIdx.x; i < n_docs; i += 256) {
    const int v = dcnt[i];
    const int it = (v + kMsQ - 1) / kMsQ;
    if (!it) continue;
    const long long t_lo = doc_ptr[i];
    const int len = (int)(doc_ptr[i + 1] - t_lo);
    const int tiles = (len + 31) >> 5;
    const int at = atomicAdd(&bpos[15 - (tiles < 15 ? tiles : 15)], it);
    for (int c = 0; c < it; ++c)
      items[at + c] = MsItem{(int)i, c * kMsQ, v - c * kMsQ < kMsQ ? v - c * kMsQ : kMsQ, len, t_lo, 0};
  }
}

template <int NBUF>
__global__ __launch_bounds__(kMsQ * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void maxsim_rescore_ring_kernel(
    const unsigned char* __restrict__ img, long n_docs, const unsigned char* __restrict__ img_q,
    const float* __restrict__ unscale_q, int q_len, float unscale_d, const MsItem* __restrict__ item_tab,
    const int* __restrict__ n_items, const int* __restrict__ dlist /*[n_docs][nq]: a document's queries*/, int nq,
    float* __restrict__ exact /*[nq, n_docs]*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // [NBUF][32 * 512]
  constexpr int kStage = 32 * 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  long poff[2];  // DMA role of this wave: pieces 2 wave, 2 wave + 1 of a tile (maxsim_scores_ring_kernel)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int prow = 2 * (2 * wave + u) + (lane >> 5);
    poff[u] = (long)prow * 512 + (((lane & 31) ^ (prow & 15)) << 4);
  }
  int foff[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) foff[st] = ms_tile_off(r32, 2 * st + h);
  const int items = *n_items;
  const int stride = (int)gridDim.x;
  auto desc_of = [&](int it) { return it < items ? item_tab[it] : MsItem{0, 0, 0, 0, 0, 0}; };
  // The block's items (blockIdx.x, + gridDim.x, ...) are ONE stream of tiles through the ring: the producer cursor runs up
  // to NBUF - 1 tiles ahead of the consumer ACROSS item boundaries (a document has ~5 tiles: restarting the ring per item
  // exposed a memory latency per item and left the ring mostly empty).  Consumer look-ahead: the descriptor of the item
  // after next and this wave's query of the next item are requested while the current item is multiplied.
  MsItem c_cur = desc_of(blockIdx.x), c_nxt = desc_of(blockIdx.x + stride);
  int qi_c = wave < c_cur.cnt ? dlist[(size_t)c_cur.doc * nq + c_cur.p0 + wave] : 0;
  MsItem p_cur = c_cur, p_nxt = c_nxt;
  int p_item = blockIdx.x, p_tile = 0;
  int issued = 0, done = 0;
  auto produce = [&]() {
    while (p_item < items && p_tile >= ((p_cur.len + 31) >> 5)) {  // the producer moves on to its next item
      p_item += stride;
      p_cur = p_nxt;
      p_nxt = desc_of(p_item + stride);
      p_tile = 0;
    }
    if (p_item >= items) return;
    const unsigned char* src = img + (size_t)(p_cur.t_lo + 32 * p_tile) * 512;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds(AMDR_MS_GPTR(src + poff[u]),
                                       AMDR_MS_LPTR(ring + (issued % NBUF) * kStage + (2 * wave + u) * 1024), 16, 0, 0);
    ++issued;
    ++p_tile;
  };
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) produce();
  for (int item = blockIdx.x; item < items; item += stride) {
    const bool live = wave < c_cur.cnt;
    const int qi = qi_c, len = c_cur.len, ntiles = (c_cur.len + 31) >> 5;
    ms8h qh[8], ql[8];
    ms_load_query_img(img_q, qi, r32, h, qh, ql);  // (a dead wave reads query 0's: its result is never stored)
    const float unscale = unscale_q[qi] * unscale_d;
    // vmcnt counts in issue order: with the fragments (the youngest requests) in, every DMA issued so far has landed
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    const int safe = issued;  // tiles below this index need no further wait by this wave
    const MsItem nn = desc_of(item + 2 * stride);
    const int qi_n = wave < c_nxt.cnt ? dlist[(size_t)c_nxt.doc * nq + c_nxt.p0 + wave] : 0;
    float best = -FLT_MAX;
    for (int t = 0; t < ntiles; ++t) {
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragment reads of the previous tile
      if (done >= safe) {
        // this wave's pieces of tile `done`: landed once at most 2 x (tiles issued behind it) requests are outstanding
        // (other loads issued meanwhile only make the wait stricter)
        const int behind = issued - done - 1;
        if (behind >= 3) {
          __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
        } else if (behind == 2) {
          __builtin_amdgcn_s_waitcnt(0x0F74);
        } else if (behind == 1) {
          __builtin_amdgcn_s_waitcnt(0x0F72);
        } else {
          __builtin_amdgcn_s_waitcnt(0x0F70);
        }
      }
      __builtin_amdgcn_s_barrier();  // everybody's pieces of tile `done` are in; tile done - 1 has been read by all
      asm volatile("" ::: "memory");
      produce();                     // into the stage of tile done - 1
      if (live) {  // wave-uniform: a wave without a query of this item (a document's last, partly filled item) only moves tiles
        const unsigned char* tile = ring + (done % NBUF) * kStage;
        ms8h ah[8], al[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const unsigned char* fp = tile + foff[st];
          ah[st] = *reinterpret_cast<const ms8h*>(fp);
          al[st] = *reinterpret_cast<const ms8h*>(fp + 256);
        }
        ms_tile_h(ah, al, qh, ql, h, len - 32 * t, best);
      }
      ++done;
    }
    const float total = ms_finish_h(best, r32, h, q_len, unscale);
    if (live && lane == 0) exact[(size_t)qi * n_docs + c_cur.doc] = total;
    c_cur = c_nxt;
    c_nxt = nn;
    qi_c = qi_n;
  }
}

// a query whose candidate list overflowed: every document, full form (rare: mass near-ties at the cut)
__global__ __launch_bounds__(256) void maxsim_overflow_kernel(const unsigned char* __restrict__ img,
                                                              const long long* __restrict__ doc_ptr, long n_docs,
                                                              const float* __restrict__ Q, int q_len, float unscale_d,
                                                              const int* __restrict__ overflow, float* __restrict__ exact) {
  extern __shared__ __attribute__((aligned(16))) unsigned char stages[];
  const int q = blockIdx.x;
  if (!overflow[q]) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ms8h qh[8], ql[8];
  float unscale;
  ms_load_query_h(Q + (size_t)q * q_len * kDim, q_len, true, lane & 31, lane >> 5, qh, ql, unscale);
  for (long doc = wave; doc < n_docs; doc += kMsWaves) {
    const float total = ms_exact_doc_lds(img, doc_ptr, doc, qh, ql, lane, q_len, unscale * unscale_d, stages + wave * 16384);
    if (lane == 0) exact[(size_t)q * n_docs + doc] = total;
  }
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

// The two-pass top-k's last step, one wave per query: only a query's candidates (<= cap, 64 for k <= 32) carry a
// re-scored value, so the k best are found among THEM — one register sort — instead of scanning the n_docs-long row that
// the select kernel had to fill with -FLT_MAX first (rowscores_topk_kernel: 18 us per 1 168 UCC-en queries, + 591 stores per
// query in the select kernel).  A query whose candidate list overflowed was re-scored in full (maxsim_overflow_kernel):
// its whole row is ranked.  Same keys (score, lower id first), same result.
__global__ __launch_bounds__(256) void maxsim_final_topk_kernel(const float* __restrict__ exact, long n_docs, int nq,
                                                                 const int* __restrict__ cand, const int* __restrict__ cnt,
                                                                 const int* __restrict__ overflow, int cap, int k,
                                                                 int cap_sel, float* __restrict__ out_scores,
                                                                 long long* __restrict__ out_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= nq) return;
  const float* row = exact + (size_t)q * n_docs;
  const bool whole = overflow[q] != 0;
  const int n = whole ? 0 : cnt[q];
  if (!whole && cap <= 64 && k <= 64) {
    C32 c = C32::pad();
    if (lane < n) {
      const int d = cand[(size_t)q * cap + lane];
      c = C32::make(row[d], (u32)d);
    }
    c = wave_sort64_desc(c, lane);
    if (lane < k) {
      const bool v = lane < n;
      out_scores[(size_t)q * k + lane] = v ? c.score() : -FLT_MAX;
      out_ids[(size_t)q * k + lane] = v ? c.id() : -1ll;
    }
    return;
  }
  WaveTopK<C32> tk;
  tk.init(reinterpret_cast<C32*>(smem) + (size_t)wave * cap_sel, cap_sel, k);
  const long total = whole ? n_docs : (long)n;
  for (long base = 0; base < total; base += 64) {
    const long i = base + lane;
    const bool v = i < total;
    long d = 0;
    if (v) d = whole ? i : (long)cand[(size_t)q * cap + i];
    tk.push_lanes(v ? C32::make(row[d], (u32)d) : C32::pad(), v, lane);
  }
  tk.finalize(lane);
  for (int j = lane; j < k; j += 64) {
    const bool v = j < tk.cnt;
    const C32 c = v ? tk.buf[j] : C32::pad();
    out_scores[(size_t)q * k + j] = v ? c.score() : -FLT_MAX;
    out_ids[(size_t)q * k + j] = v ? c.id() : -1ll;
  }
}

}  // namespace amdr

using namespace amdr;

struct amdr_maxsim {
  int device = 0;
  int64_t n_docs = 0, n_tokens = 0;
  float* D = nullptr;
  unsigned char* img = nullptr;  // [hi 128 x fp16 | lo 128 x fp16] per token, scaled by d_scale (split-fp16 form)
  unsigned char* img_hi = nullptr;  // [hi 128 x fp16] per token: first pass of the two-pass top-k
  float d_scale = 1.f;           // power of two; img == nullptr: the store is not finite -> fp32-input form only
  float d_norm_max = 0.f;        // largest token L2 norm (error bound of the first pass)
  long long* doc_ptr = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  DevBuf full[2], qbuf, sbuf, ibuf;  // full[0]: "_device" calls, full[1]: host-pointer calls (see dense.hip)
};

namespace {

bool ms_half(const amdr_maxsim* h) {
  const char* pin = getenv("AMDR_MAXSIM_F16X3");
  return h->img != nullptr && !(pin && pin[0] == '0');  // split-fp16 MFMA form (default) / fp32-input form
}
// the two-pass top-k: batches on the split-fp16 form, k small against the corpus (AMDR_MAXSIM_TWOPASS=0 pins one pass)
bool ms_two_pass(const amdr_maxsim* h, int nq, int k, bool want_topk) {
  const char* pin = getenv("AMDR_MAXSIM_TWOPASS");
  if (pin && pin[0] == '0') return false;
  return want_topk && ms_half(h) && nq >= kMsQ && h->img_hi != nullptr && (int64_t)4 * k <= h->n_docs;
}
int ms_cand_cap(int k) {
  const int c = next_pow2(2 * k);
  return c < 64 ? 64 : c;
}
// workspace of one call: the score rows [nq, n_docs]; two-pass: the first-pass rows, the re-scored rows, the candidate lists
size_t ms_workspace_bytes(const amdr_maxsim* h, int nq, int k, bool want_topk) {
  const size_t rows = ((size_t)nq * h->n_docs * sizeof(float) + 255) / 256 * 256;
  if (!ms_two_pass(h, nq, k, want_topk)) return rows;
  // three row blocks (first-pass scores, re-scored scores, the per-document query lists of the re-scoring pass) + the
  // candidate lists [nq * cap], counters, the item table (<= n_docs + pairs / 8 descriptors of 32 bytes); an upper bound
  return 3 * rows + ((size_t)nq * ms_cand_cap(k) + 3 * (size_t)nq + 1) * sizeof(int) + 256 +  // (3rd: dlist [n_docs][nq])
         ((size_t)nq * ms_cand_cap(k) + 4 * (size_t)h->n_docs + 8) * sizeof(int) +
         ((size_t)h->n_docs + (size_t)nq * ms_cand_cap(k) / kMsQ + 8) * sizeof(MsItem) +
         (size_t)nq * (32 * 512 + 2 * sizeof(float)) + 512;  // + the split image of the queries, their scales, norm sums
}

int ms_run(amdr_maxsim* h, const float* Q_dev, int nq, int q_len, int k, float* full_dev, float* scores_dev,
           int64_t* ids_dev, hipStream_t st) {
  const bool half = ms_half(h);
  const float unscale_d = 1.f / h->d_scale;
  const bool batch = nq >= kMsQ;  // batches: document tiles shared by 8 queries through LDS
  if (ms_two_pass(h, nq, k, scores_dev != nullptr)) {
    const size_t rows = ((size_t)nq * h->n_docs * sizeof(float) + 255) / 256 * 256;
    unsigned char* wsb = reinterpret_cast<unsigned char*>(full_dev);
    float* approx = full_dev;
    float* exact = reinterpret_cast<float*>(wsb + rows);
    const int cap = ms_cand_cap(k);
    int* dlist = reinterpret_cast<int*>(wsb + 2 * rows);  // [n_docs][nq] the queries a document is a candidate of
    int* cand = reinterpret_cast<int*>(wsb + 3 * rows);
    int* cnt = cand + (size_t)nq * cap;
    int* ovf = cnt + nq;
    const int cap_sel = topk_cap(k);
    int* off = ovf + nq;                      // [nq + 1] (round-3 form)
    int* dcnt = off + nq + 1;                 // round 4: [n_docs] candidates per document
    int* ioff = dcnt + h->n_docs;             // [1] items of the re-scoring pass
    MsItem* items = reinterpret_cast<MsItem*>(((uintptr_t)(ioff + 1) + 31) & ~(uintptr_t)31);
    unsigned char* img_q = reinterpret_cast<unsigned char*>(
        ((uintptr_t)(items + h->n_docs + (size_t)nq * cap / kMsQ + 8) + 255) & ~(uintptr_t)255);
    float* unscale_q = reinterpret_cast<float*>(img_q + (size_t)nq * 32 * 512);
    float* nsum_q = unscale_q + nq;
    const char* rs = getenv("AMDR_MAXSIM_RESCORE");  // "0": one wave per pair (the round-3 form; A/B, tests)
    const bool by_doc = !(rs && rs[0] == '0');
    const char* fc = getenv("AMDR_MAXSIM_FINAL");  // "0": rank the whole re-scored rows (rowscores_topk_kernel; A/B, tests)
    const bool final_cand = !(fc && fc[0] == '0');
    // documents per block of pass 1 (round 4, scripts/ab_maxsim_env.py — variants interleaved in one process, channel ms):
    // UCC-en 16 / 24 / 29 / 32: 0.891 / 0.884 / 0.897 / 0.903, Civil-Code-zh 16 / 24 / 32 / 48 / 64: 1.161 / 1.161 / 1.163 /
    // 1.101* / 1.095* (*another box: 32 = 1.090).  Whole rounds of the chip's block slots (29 documents: 3 066 blocks = 3.99
    // rounds of 768, against 3.6 at 32) bought nothing, nor did 2 or 4 ring stages (AMDR_MAXSIM_RING1; 5 or 2 blocks per
    // CU instead of 3: -0.5 % / +2 %): the pass sits on a plateau that scheduling does not move.
    const char* dpb = getenv("AMDR_MAXSIM_DOCS");
    long docs = dpb && atoi(dpb) > 0 ? atoi(dpb) : 24;
    while (ceil_div(h->n_docs, docs) > 65535) docs *= 2;
    if (by_doc)
      hipLaunchKernelGGL(maxsim_split_queries_kernel, dim3(ceil_div(nq, 4)), dim3(256), 0, st, Q_dev, nq, q_len, img_q,
                         unscale_q, nsum_q);
    const char* h2 = getenv("AMDR_MAXSIM_HI2");  // "0": one query per wave (A/B)
    if (h2 && h2[0] == '0') {
      AMDR_HIP(hipFuncSetAttribute((const void*)maxsim_hi_ring_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   4 * 16384));
      hipLaunchKernelGGL((maxsim_hi_ring_kernel<4>), dim3(ceil_div(nq, kMsQ), ceil_div(h->n_docs, docs)), dim3(kMsQ * 64),
                         4 * 16384, st, h->img_hi, h->doc_ptr, (long)h->n_docs, (int)docs, Q_dev, nq, q_len, approx,
                         unscale_d);
    } else {
      const char* ps = getenv("AMDR_MAXSIM_PRESPLIT");  // "0": every block of pass 1 splits its queries itself (A/B)
      const bool presplit = by_doc && !(ps && ps[0] == '0');
      const char* r1 = getenv("AMDR_MAXSIM_RING1");  // LDS stages of pass 1 (2 / 3 / 4: 5 / 3 / 2 blocks per CU)
      const int nb1 = r1 ? atoi(r1) : 3;
#define AMDR_MS_PASS1(NB)                                                                                              \
  do {                                                                                                                 \
    AMDR_HIP(hipFuncSetAttribute((const void*)maxsim_hi2_ring_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                 NB * 16384));                                                                         \
    hipLaunchKernelGGL((maxsim_hi2_ring_kernel<NB>), dim3(ceil_div(nq, 2 * kMsQ2), ceil_div(h->n_docs, docs)),        \
                       dim3(kMsQ2 * 64), NB * 16384, st, h->img_hi, h->doc_ptr, (long)h->n_docs, (int)docs, Q_dev, nq, \
                       q_len, approx, unscale_d, presplit ? img_q : (const unsigned char*)nullptr, unscale_q);         \
  } while (0)
      if (nb1 == 2) AMDR_MS_PASS1(2);
      else if (nb1 == 4) AMDR_MS_PASS1(4);
      else AMDR_MS_PASS1(3);
#undef AMDR_MS_PASS1
    }
    if (by_doc) AMDR_HIP(hipMemsetAsync(dcnt, 0, (size_t)h->n_docs * sizeof(int), st));
    hipLaunchKernelGGL(maxsim_select_kernel, dim3(nq), dim3(64), (size_t)cap_sel * sizeof(C32), st, approx,
                       (long)h->n_docs, Q_dev, q_len, k, cap_sel, h->d_norm_max, unscale_d, cap, exact, cand, cnt, ovf,
                       by_doc ? dcnt : (int*)nullptr, dlist, nq, by_doc ? nsum_q : (const float*)nullptr, unscale_q,
                       final_cand ? 0 : 1);
    constexpr int kPairLds = kMsWaves * 16384;
    AMDR_HIP(hipFuncSetAttribute((const void*)maxsim_overflow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
    int dev = 0, cus = 256;
    AMDR_HIP(hipGetDevice(&dev));
    AMDR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (by_doc) {
      hipLaunchKernelGGL(maxsim_items_kernel, dim3(1), dim3(256), 0, st, dcnt, (long)h->n_docs, h->doc_ptr, ioff, items);
      const char* e_rr = getenv("AMDR_MAXSIM_RESCORE_RING");
      const char* e_rb = getenv("AMDR_MAXSIM_RESCORE_BLOCKS");
      // Ring depth and blocks per CU of the re-scoring pass (same process, interleaved, 1 168 UCC-en queries, whole channel):
      // 4 stages x 2 blocks per CU (64 KB of LDS each: 4 waves per SIMD) 0.8916 ms; 3 x 3: 0.8903; 2 x 4 (8 waves per SIMD):
      // 0.8837; 2 stages with 6 / 8 blocks per CU in the grid (the items then outnumber the blocks by little: the hardware
      // deals them) 0.8815 / 0.8793.  What the deeper ring bought inside a block, twice the resident waves buy across
      // blocks: the per-item latencies (descriptor -> query list -> query fragments -> first tile) overlap another
      // block's products.  Civil-Code-zh: 1.1017 -> 1.0947.
      const int rr = e_rr ? atoi(e_rr) : 2;
      const int rb = e_rb ? atoi(e_rb) : (rr == 3 ? 3 : rr == 2 ? 8 : 2);
#define AMDR_MS_RESCORE(NB)                                                                                              \
  do {                                                                                                                   \
    AMDR_HIP(hipFuncSetAttribute((const void*)maxsim_rescore_ring_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 NB * 16384));                                                                           \
    hipLaunchKernelGGL((maxsim_rescore_ring_kernel<NB>), dim3((rb >= 1 && rb <= 8 ? rb : 2) * cus), dim3(kMsQ * 64),     \
                       NB * 16384, st, h->img, (long)h->n_docs, img_q, unscale_q, q_len, unscale_d, items, ioff, dlist,  \
                       nq, exact);                                                                                       \
  } while (0)
      if (rr == 3) AMDR_MS_RESCORE(3);
      else if (rr == 2) AMDR_MS_RESCORE(2);
      else AMDR_MS_RESCORE(4);
#undef AMDR_MS_RESCORE
    } else {
      AMDR_HIP(hipFuncSetAttribute((const void*)maxsim_rescore_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
      hipLaunchKernelGGL(maxsim_offsets_kernel, dim3(1), dim3(256), 0, st, cnt, nq, off);
      hipLaunchKernelGGL(maxsim_rescore_kernel, dim3(2 * cus), dim3(256), kPairLds, st, h->img, h->doc_ptr, (long)h->n_docs,
                         Q_dev, nq, q_len, unscale_d, cand, off, cap, exact);
    }
    hipLaunchKernelGGL(maxsim_overflow_kernel, dim3(nq), dim3(256), kPairLds, st, h->img, h->doc_ptr, (long)h->n_docs,
                       Q_dev, q_len, unscale_d, ovf, exact);
    AMDR_HIP(hipGetLastError());
    if (final_cand) {
      hipLaunchKernelGGL(maxsim_final_topk_kernel, dim3(ceil_div(nq, 4)), dim3(256), (size_t)4 * cap_sel * sizeof(C32), st,
                         exact, (long)h->n_docs, nq, cand, cnt, ovf, cap, k, cap_sel, scores_dev, (long long*)ids_dev);
      AMDR_HIP(hipGetLastError());
      return AMDR_OK;
    }
    full_dev = exact;  // the final top-k ranks the re-scored rows
  } else if (batch && half) {
    const char* rg = getenv("AMDR_MAXSIM_RING");  // LDS stages (2 / 3 / 4 / 6; measured best: 4)
    const int nbuf = rg ? atoi(rg) : 4;
    const char* dpb = getenv("AMDR_MAXSIM_DOCS");  // documents per block (measured best: 16)
    long docs = dpb && atoi(dpb) > 0 ? atoi(dpb) : 16;
    while (ceil_div(h->n_docs, docs) > 65535) docs *= 2;
    dim3 grid(ceil_div(nq, kMsQ), ceil_div(h->n_docs, docs));
#define AMDR_MS_RING(NB)                                                                                             \
  {                                                                                                                  \
    AMDR_HIP(hipFuncSetAttribute((const void*)maxsim_scores_ring_kernel<NB>,                                         \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, NB * 16384));                           \
    hipLaunchKernelGGL((maxsim_scores_ring_kernel<NB>), grid, dim3(kMsQ * 64), NB * 16384, st, h->img, h->doc_ptr,    \
                       (long)h->n_docs, (int)docs, Q_dev, nq, q_len, full_dev, unscale_d);                           \
  }
    if (nbuf == 2) {
      AMDR_MS_RING(2)
    } else if (nbuf == 3) {
      AMDR_MS_RING(3)
    } else if (nbuf == 6) {
      AMDR_MS_RING(6)
    } else {
      AMDR_MS_RING(4)
    }
#undef AMDR_MS_RING
  } else if (batch && ceil_div(h->n_docs, kMsDocs) <= 65535) {
    dim3 grid(ceil_div(nq, kMsQ), ceil_div(h->n_docs, kMsDocs));
    hipLaunchKernelGGL(maxsim_scores_blocked_kernel, grid, dim3(kMsQ * 64), 0, st, h->D, h->doc_ptr, (long)h->n_docs,
                       (long)h->n_tokens, Q_dev, nq, q_len, full_dev);
  } else {
    dim3 grid(ceil_div(h->n_docs, kMsWaves), nq);
    if (half)
      hipLaunchKernelGGL(maxsim_scores_h_kernel, grid, dim3(256), 0, st, h->img, h->doc_ptr, (long)h->n_docs, Q_dev,
                         q_len, full_dev, unscale_d);
    else
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
  // the split-fp16 image: the store's largest |component| fixes a power-of-two scale into [0.5, 1), then every
  // token row is split once (a store with a NaN / infinity keeps the fp32-input form only)
  unsigned int* mx = nullptr;
  unsigned int mbits = 0;
  if (e == hipSuccess) e = hipMalloc((void**)&mx, sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemset(mx, 0, sizeof(unsigned int));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ms_absmax_kernel, dim3(1024), dim3(256), 0, 0, h->D, (long)(nt * dim), mx);
    e = hipMemcpy(&mbits, mx, sizeof(unsigned int), hipMemcpyDeviceToHost);
  }
  if (mx) (void)hipFree(mx);
  if (e == hipSuccess && mbits < 0x7f800000u) {
    float m;
    memcpy(&m, &mbits, sizeof(float));
    int ex = 0;
    if (m > 0.f) (void)frexpf(m, &ex);
    h->d_scale = ldexpf(1.f, -ex);
    e = hipMalloc((void**)&h->img, (size_t)(nt + 32) * 512);  // + one tile: the last tile of the last document reads on
    if (e == hipSuccess) e = hipMemset(h->img + (size_t)nt * 512, 0, (size_t)32 * 512);
    if (e == hipSuccess) e = hipMalloc((void**)&h->img_hi, (size_t)(nt + 64) * 256);  // + one 64-token tile
    if (e == hipSuccess) e = hipMemset(h->img_hi + (size_t)nt * 256, 0, (size_t)64 * 256);
    unsigned int* nm = nullptr;
    unsigned int nbits = 0;
    if (e == hipSuccess) e = hipMalloc((void**)&nm, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(nm, 0, sizeof(unsigned int));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(ms_split_store_kernel, dim3(ceil_div(nt * 16, 256)), dim3(256), 0, 0, h->D, (long)nt, h->d_scale,
                         h->img, h->img_hi);
      hipLaunchKernelGGL(ms_tokmax_kernel, dim3(1024), dim3(256), 0, 0, h->D, (long)nt, nm);
      e = hipMemcpy(&nbits, nm, sizeof(unsigned int), hipMemcpyDeviceToHost);
      memcpy(&h->d_norm_max, &nbits, sizeof(float));
    }
    if (nm) (void)hipFree(nm);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
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

int amdr_maxsim_plan_info(const amdr_maxsim_t* h, int32_t nq, char* buf, int32_t buf_len) {
  AMDR_REQUIRE(h && buf && buf_len > 0, "maxsim_plan_info: null");
  AMDR_REQUIRE(nq >= 1, "maxsim_plan_info: nq=%d", nq);
  const char* pin = getenv("AMDR_MAXSIM_F16X3");
  const bool half = h->img != nullptr && !(pin && pin[0] == '0');
  const bool batch = nq >= kMsQ;
  if (half && batch && h->img_hi && !(getenv("AMDR_MAXSIM_TWOPASS") && getenv("AMDR_MAXSIM_TWOPASS")[0] == '0'))
    snprintf(buf, buf_len,
             "maxsim_hi2_ring_kernel split-fp16 two-pass top-k (k <= n_docs / 4): maxsim_split_queries_kernel + pass 1 hi "
             "parts only (1 x v_mfma_f32_32x32x16_f16 per block, two queries per wave) + maxsim_select_kernel + %s "
             "(hi + lo/2048, 3 MFMAs per block, candidates only) + %s; full score rows: maxsim_scores_ring_kernel",
             (getenv("AMDR_MAXSIM_RESCORE") && getenv("AMDR_MAXSIM_RESCORE")[0] == '0')
                 ? "maxsim_rescore_kernel (one wave per candidate pair)"
                 : "maxsim_rescore_ring_kernel<2> (pairs grouped by document: a block = one document x 8 of its queries; 8 blocks per CU)",
             (getenv("AMDR_MAXSIM_FINAL") && getenv("AMDR_MAXSIM_FINAL")[0] == '0')
                 ? "rowscores_topk_kernel"
                 : "maxsim_final_topk_kernel (the candidates only)");
  else if (half)
    snprintf(buf, buf_len, "%s split-fp16 (hi + lo/2048, 3 x v_mfma_f32_32x32x16_f16 per block) + rowscores_topk_kernel",
             batch ? "maxsim_scores_ring_kernel" : "maxsim_scores_h_kernel");
  else
    snprintf(buf, buf_len, "%s fp32-input (v_mfma_f32_16x16x4_f32) + rowscores_topk_kernel",
             batch ? "maxsim_scores_blocked_kernel" : "maxsim_scores_kernel");
  return AMDR_OK;
}

int amdr_maxsim_reserve(amdr_maxsim_t* h, int32_t nq_max, int32_t k_max) {
  AMDR_REQUIRE(h != nullptr, "maxsim_reserve: null handle");
  AMDR_REQUIRE(nq_max >= 1 && k_max >= 1 && k_max <= AMDR_MAX_K, "maxsim_reserve: bad sizes");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  // every call within (nq_max, k_max): the one-pass rows, or the two-pass layout at the largest candidate capacity
  size_t need = ms_workspace_bytes(h, nq_max, k_max, true);
  if (ms_half(h) && h->img_hi) {
    const size_t rows = ((size_t)nq_max * h->n_docs * sizeof(float) + 255) / 256 * 256;
    const size_t two = 2 * rows + ((size_t)nq_max * ms_cand_cap(k_max) + 3 * (size_t)nq_max + 1) * sizeof(int) + 256;
    need = two > need ? two : need;
  }
  int rc = h->full[0].ensure(need);
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
  if ((rc = h->full[0].ensure(ms_workspace_bytes(h, nq, k, true)))) return rc;
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
  if ((rc = h->full[1].ensure(ms_workspace_bytes(h, nq, k, true)))) return rc;
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
  if (h->img) (void)hipFree(h->img);
  if (h->img_hi) (void)hipFree(h->img_hi);
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
