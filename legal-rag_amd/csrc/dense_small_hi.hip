// First pass of a two-pass form of the LONG-BATCH dense channel on a SHORT corpus (the headline step: 37 376 queries x 591
// chunks x 768 dims, hybrid_retriever.py:181-189 search_dense over a batch): approximate scores of every (query, chunk)
// on the fp16 matrix instructions, 16 x the rate of the exact fp32 form that dense_panel.hip runs at 0.80 of its peak, with
// the PROVEN bound of dense_hi.hip on their distance from the exact dot product — what a second pass needs to re-score only
// the chunks within 2 eps of a query's k-th best (fuse.hip dense_hi_select_fuse_kernel; dense.hip run_search_batched takes
// the pair from 4 096 queries per launch).
//
//   amdr_dense_small_create        statistics of the chunk matrix (largest component -> power-of-two scale, largest row
//                                  norm) and its fp16 image  Xh[K slice][row][128 halves]  (rows padded to 32, zero)
//   dsh_split_queries_kernel       per query: power-of-two scale, the proven bound, fp16 image in MFMA FRAGMENT order
//                                  Qh[query tile of 32][K slice][k step][lane][8 halves] — one coalesced 1-KiB load per
//                                  fragment in the scores kernel
//   dsh_scores_kernel<NBUF>        a block = 4 waves x 2 query tiles (256 queries) x 4 chunk tiles (128 chunks); the chunk
//                                  tiles of one K slice (32 rows x 256 B = 8 KiB each) go through an LDS ring by LDS-DMA,
//                                  shared by the 8 query tiles; K slice outermost: a wave re-loads its 2 x 8 query
//                                  fragments per slice (64 VGPRs) and keeps 4 x 2 accumulator tiles (128 VGPRs) across
//                                  the 6 slices of d = 768 — 32 queries x 768 dims of fragments (192 VGPRs for ONE
//                                  tile) do not fit a wave, which is what makes this shape different from MaxSim's.
// Error bound (dense_hi.hip, same arithmetic: fp16 roundings of both scaled operands, exact products, fp32 accumulation):
//   |approx - exact| <= eps_q = [1.125 (2^-10 + 2^-22 + 2 (d + 8) 2^-24) |q'| R' + 1.125 d 2^-24] / (x_scale q_scale)
// in the units of the exact score (q' = q q_scale, R' = largest row norm x x_scale).
#include "common.hpp"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

namespace amdr {

#define AMDR_DS_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AMDR_DS_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef _Float16 ds8h __attribute__((ext_vector_type(8)));
typedef float ds4f __attribute__((ext_vector_type(4)));
typedef float dsf16 __attribute__((ext_vector_type(16)));

constexpr int kDsWaves = 4;  // waves per block, two 32-query tiles each
constexpr int kDsG = 4;      // 32-row chunk tiles per block
constexpr int kDsStage = 32 * 256;

__device__ __forceinline__ int ds_hi_off(int row, int slot) { return row * 256 + ((slot ^ (row & 15)) << 4); }

// ---- statistics of the chunk matrix: out[0] = largest |component|, out[1] = largest row norm (as uint bit patterns of
// non-negative floats: ordered like the floats; a NaN poisons both through 0x7fc00000 > every finite pattern)
__global__ __launch_bounds__(256) void dsh_stats_kernel(const float* __restrict__ X, long n, int d, unsigned int* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= n) return;
  float amax = 0.f, ss = 0.f;
  bool nan = false;
  for (int j = lane * 4; j < d; j += 256) {
    const ds4f v = *reinterpret_cast<const ds4f*>(X + (size_t)row * d + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      nan |= v[e] != v[e];
      amax = fmaxf(amax, fabsf(v[e]));
      ss += v[e] * v[e];
    }
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) {
    amax = fmaxf(amax, __shfl_xor(amax, sft));
    ss += __shfl_xor(ss, sft);
  }
  nan = __any(nan);
  if (lane == 0) {
    atomicMax(out, nan ? 0x7fc00000u : __float_as_uint(amax));
    atomicMax(out + 1, nan ? 0x7fc00000u : __float_as_uint(sqrtf(ss)));
  }
}

// ---- the chunk image: Xh[slice][row][slot of 8 halves] = fp16(x * x_scale), one thread per (row, 8 components)
__global__ __launch_bounds__(256) void dsh_image_kernel(const float* __restrict__ X, long n, int d, float x_scale,
                                                        unsigned char* __restrict__ Xh, long n_pad) {
  const int groups = d / 8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * groups) return;
  const long row = idx / groups;
  const int g = (int)(idx - row * groups);
  const ds4f v0 = *reinterpret_cast<const ds4f*>(X + (size_t)row * d + 8 * g);
  const ds4f v1 = *reinterpret_cast<const ds4f*>(X + (size_t)row * d + 8 * g + 4);
  ds8h y;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y[e] = (_Float16)(v0[e] * x_scale);
    y[4 + e] = (_Float16)(v1[e] * x_scale);
  }
  *reinterpret_cast<ds8h*>(Xh + ((size_t)(g >> 4) * n_pad + row) * 256 + (size_t)(g & 15) * 16) = y;
}

// ---- the queries: a wave takes EIGHT consecutive queries, eight lanes each (lane = 8 j + s: query j, groups s, s + 8, ...
// of 8 components): for one group index the eight queries' 16-byte pieces are neighbours in the fragment — every store
// instruction writes whole 128-byte lines (one query per wave wrote 16 bytes per line and instruction: 44 us per 37 376
// queries for 172 MB, under half of what HBM takes).  Also the per-query bound, in the units of the exact score.
__global__ __launch_bounds__(256) void dsh_split_queries_kernel(const float* __restrict__ Q, int nq, int d,
                                                                unsigned char* __restrict__ Qh, float* __restrict__ q_unscale,
                                                                float r_scaled, float x_unscale, float* __restrict__ eps) {
  const int lane = threadIdx.x & 63, jq = lane >> 3, sub = lane & 7;
  const int q = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + jq;
  const bool live = q < nq;
  const int per = d >> 6, ks = d >> 7;  // groups per lane (d / 8 groups over 8 lanes)
  float x[16][8];
  float amax = 0.f;
  bool nan = false;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < per && live) {
      const float* p = Q + (size_t)q * d + 8 * (sub + 8 * i);
      const ds4f v0 = *reinterpret_cast<const ds4f*>(p);
      const ds4f v1 = *reinterpret_cast<const ds4f*>(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[i][e] = v0[e], x[i][4 + e] = v1[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[i][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      nan |= x[i][e] != x[i][e];
      amax = fmaxf(amax, fabsf(x[i][e]));
    }
  }
#pragma unroll
  for (int sft = 1; sft < 8; sft <<= 1) {
    amax = fmaxf(amax, __shfl_xor(amax, sft));
    nan |= __shfl_xor((int)nan, sft) != 0;
  }
  int e2 = 0;
  if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e2);  // amax = f 2^e2, f in [0.5, 1)
  const float sc = ldexpf(1.f, -e2);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = x[i][e] * sc;
      ss += v * v;
    }
#pragma unroll
  for (int sft = 1; sft < 8; sft <<= 1) ss += __shfl_xor(ss, sft);
  if (!live) return;
  const int tile = q >> 5, r32 = q & 31;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i >= per) continue;
    const int g = sub + 8 * i;
    const int kc = g >> 4, slot = g & 15, st = slot >> 1, hh = slot & 1;
    ds8h y;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (_Float16)(x[i][e] * sc);
    *reinterpret_cast<ds8h*>(Qh + ((((size_t)tile * ks + kc) * 8 + st) * 64 + (r32 + 32 * hh)) * 16) = y;
  }
  if (sub == 0) {
    const bool bad = nan || !(amax <= FLT_MAX) || e2 > 100 || e2 < -100;
    const float us = ldexpf(1.f, e2);
    q_unscale[q] = us;
    const float rel = 1.125f * (9.765625e-4f + 2.4e-7f + 2.f * (float)(d + 8) * 5.9604645e-8f);
    const float e_q = (rel * sqrtf(ss) * r_scaled + 1.125f * (float)d * 5.9604645e-8f) * x_unscale * us;
    if (eps) eps[q] = bad ? __uint_as_float(0x7fc00000u) : e_q;  // NaN: no bound for this query
  }
}

// ---- the scores
template <int NBUF>
__global__ __launch_bounds__(kDsWaves * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void dsh_scores_kernel(
    const unsigned char* __restrict__ Xh, long n_pad, int n_tiles, int ks, const unsigned char* __restrict__ Qh, int nq,
    const float* __restrict__ q_unscale, float x_unscale, float* __restrict__ S, long ldS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // [NBUF][32 * 256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int n_qtiles = (nq + 31) >> 5;
  // Block -> (query group, chunk group), XCD-aware: consecutive workgroup ids go round the 8 XCDs, each with its own L2, and
  // the chunk groups of ONE query group re-read the same 384 KB of query fragments — so they take consecutive slots of the
  // SAME XCD (b % 8) and the fragments come from that L2 after the first of them.  In (x = query group, y = chunk group)
  // order the fabric carried the query image once per chunk group: 297 MB of reads per 37 376 x 591 launch for 57 MB.
  const int n_groups = (n_tiles + kDsG - 1) / kDsG;
  const int b = blockIdx.x, slot = b >> 3;
  const int qg = (slot / n_groups) * 8 + (b & 7), yg = slot % n_groups;
  const int ta = (qg * kDsWaves + wave) * 2, tb = ta + 1;
  if (qg * kDsWaves * 2 >= n_qtiles) return;  // (block-uniform: the grid is rounded up to 8 query groups)
  const int la = ta < n_qtiles ? ta : 0, lb = tb < n_qtiles ? tb : 0;  // (a dead tile loads tile 0's fragments, stores nothing)
  const int t0 = yg * kDsG;
  const int g = n_tiles - t0 < kDsG ? n_tiles - t0 : kDsG;  // block-uniform, >= 1

  // DMA role: pieces 2 wave, 2 wave + 1 of a tile's 8 (1 KiB = 4 rows of 256 B); lane l: row + (l >> 4), PHYSICAL slot
  // l & 15, which holds logical slot ^ (row & 15) (maxsim.hip, the hi-only ring)
  long poff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int prow = 4 * (2 * wave + u) + (lane >> 4);
    poff[u] = (long)prow * 256 + (((lane & 15) ^ (prow & 15)) << 4);
  }
  int foff[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) foff[st] = ds_hi_off(r32, 2 * st + h);

  // the stream of tiles through the ring: K slice major, the block's g chunk tiles inside
  int p_kc = 0, p_j = 0;  // producer cursor
  auto issue = [&](int stage) {
    const unsigned char* src = Xh + ((size_t)p_kc * n_pad + (size_t)32 * (t0 + p_j)) * 256;  // wave-uniform
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds(AMDR_DS_GPTR(src + poff[u]),
                                       AMDR_DS_LPTR(ring + stage * kDsStage + (2 * wave + u) * 1024), 16, 0, 0);
    if (++p_j == g) p_j = 0, ++p_kc;
  };
  int issued = 0, done = 0;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) {
    if (p_kc < ks) {
      issue(issued % NBUF);
      ++issued;
    }
  }
  dsf16 acc[kDsG][2];
#pragma unroll
  for (int j = 0; j < kDsG; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][0][i] = acc[j][1][i] = 0.f;

  for (int kc = 0; kc < ks; ++kc) {
    // this slice's query fragments: 2 x 8 coalesced 1-KiB loads; drained at once (they are needed now), which also lands
    // every tile issued so far
    ds8h qa[8], qb[8];
    const unsigned char* pa = Qh + (((size_t)la * ks + kc) * 8) * 1024 + (size_t)lane * 16;
    const unsigned char* pb = Qh + (((size_t)lb * ks + kc) * 8) * 1024 + (size_t)lane * 16;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      qa[st] = *reinterpret_cast<const ds8h*>(pa + st * 1024);
      qb[st] = *reinterpret_cast<const ds8h*>(pb + st * 1024);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < kDsG; ++j) {
      if (j < g) {  // block-uniform
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragment reads of the previous tile
        const int behind = issued - done - 1;  // tiles issued behind this one: 2 loads each may still be in flight
        if (behind >= 3) {
          __builtin_amdgcn_s_waitcnt(0x0F76);
        } else if (behind == 2) {
          __builtin_amdgcn_s_waitcnt(0x0F74);
        } else if (behind == 1) {
          __builtin_amdgcn_s_waitcnt(0x0F72);
        } else {
          __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        __builtin_amdgcn_s_barrier();  // everybody's pieces of tile `done` are in; tile done - 1 has been read by all
        asm volatile("" ::: "memory");
        if (p_kc < ks) {
          issue(issued % NBUF);
          ++issued;
        }
        const unsigned char* tile = ring + (done % NBUF) * kDsStage;
        ds8h a[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) a[st] = *reinterpret_cast<const ds8h*>(tile + foff[st]);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st], qa[st], acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st], qb[st], acc[j][1], 0, 0, 0);
        }
        ++done;
      }
    }
  }
  // accumulator element i of lane (r32, h): chunk row 32 (t0 + j) + (i & 3) + 8 (i >> 2) + 4 h, query 32 tile + r32
#pragma unroll
  for (int ab = 0; ab < 2; ++ab) {
    const int tq = ab == 0 ? ta : tb;
    const int q = tq * 32 + r32;
    if (tq >= n_qtiles || q >= nq) continue;
    const float us = x_unscale * q_unscale[q];  // both powers of two: exact
    float* row = S + (size_t)q * ldS + (size_t)32 * t0 + 4 * h;
#pragma unroll
    for (int j = 0; j < kDsG; ++j) {
      if (j >= g) continue;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        ds4f v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[j][ab][4 * m + e] * us;
        *reinterpret_cast<ds4f*>(row + 32 * j + 8 * m) = v;
      }
    }
  }
}

}  // namespace amdr

using namespace amdr;

struct amdr_dense_small {
  int device = 0;
  const float* X = nullptr;
  int64_t n = 0, n_pad = 0;
  int d = 0;
  float x_scale = 1.f, row_norm_max = 0.f;
  bool ok = false;
  DevBuf img, ws;
  std::mutex mu;
};

namespace amdr {
static size_t dsh_ws_bytes(int d, int nq, size_t* qh_bytes) {
  const size_t qh = ((size_t)((nq + 31) / 32) * (d / 128) * 8 * 1024 + 255) / 256 * 256;
  if (qh_bytes) *qh_bytes = qh;
  return qh + (size_t)nq * sizeof(float) + 256;
}
int dense_small_reserve(amdr_dense_small_t* h, int nq_max) {
  std::lock_guard<std::mutex> g(h->mu);
  return h->ws.ensure(dsh_ws_bytes(h->d, nq_max, nullptr));
}
bool dense_small_usable(const amdr_dense_small_t* h) { return h && h->ok; }
// (the caller has set the device; X stays the caller's)
int dense_small_create_from(int device, const float* X, int64_t n, int d, amdr_dense_small_t** out) {
  *out = nullptr;
  AMDR_REQUIRE(n >= 1 && d >= 128 && d <= 1024 && d % 128 == 0,
               "dense_small_create: needs >= 1 row and d a multiple of 128 in [128, 1024] (d=%d)", d);
  amdr_dense_small* h = new (std::nothrow) amdr_dense_small();
  if (!h) return fail(AMDR_ENOMEM, "dense_small_create: host alloc");
  int rc = AMDR_OK;
  h->device = device;
  h->X = X;
  h->n = n;
  h->d = d;
  h->n_pad = (n + 31) / 32 * 32;
  unsigned int* st = nullptr;
  unsigned int host[2] = {0u, 0u};
  const size_t img_bytes = (size_t)(h->d / 128) * h->n_pad * 256;
  rc = h->img.ensure(img_bytes);
  if (!rc && hipMalloc((void**)&st, 2 * sizeof(unsigned int)) != hipSuccess) rc = fail(AMDR_EHIP, "dense_small_create: alloc");
  if (!rc) {
    (void)hipMemset(st, 0, 2 * sizeof(unsigned int));
    hipLaunchKernelGGL(dsh_stats_kernel, dim3(ceil_div(h->n, 4)), dim3(256), 0, nullptr, h->X, (long)h->n, h->d, st);
    if (hipMemcpy(host, st, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(AMDR_EHIP, "dense_small_create: stats");
  }
  if (st) (void)hipFree(st);
  if (!rc) {
    float amax, rmax;
    memcpy(&amax, &host[0], 4);
    memcpy(&rmax, &host[1], 4);
    int e = 0;
    if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e);
    h->x_scale = ldexpf(1.f, -e);
    h->row_norm_max = rmax;
    h->ok = amax <= FLT_MAX && rmax <= FLT_MAX && e > -100 && e < 100;  // (NaN compares false)
    (void)hipMemset(h->img.p, 0, img_bytes);
    hipLaunchKernelGGL(dsh_image_kernel, dim3(ceil_div(h->n * (h->d / 8), 256)), dim3(256), 0, nullptr, h->X, (long)h->n, h->d,
                       h->x_scale, h->img.as<unsigned char>(), (long)h->n_pad);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(AMDR_EHIP, "dense_small_create: image");
  }
  if (rc) {
    h->img.release();
    delete h;
    return rc;
  }
  *out = h;
  return AMDR_OK;
}
}  // namespace amdr

extern "C" {

int amdr_dense_small_create(amdr_dense_t* dense, amdr_dense_small_t** out) {
  AMDR_REQUIRE(dense && out, "dense_small_create: null");
  *out = nullptr;
  DenseRaw raw;
  int rc;
  std::lock_guard<std::mutex> g(dense_mutex(dense));
  AMDR_HIP(hipSetDevice(dense_device_of(dense)));
  if ((rc = dense_small_raw(dense, 1, &raw))) return rc;
  return dense_small_create_from(dense_device_of(dense), raw.X, raw.n, raw.d, out);
}

int amdr_dense_small_destroy(amdr_dense_small_t* h) {
  if (!h) return AMDR_OK;
  (void)hipSetDevice(h->device);
  h->img.release();
  h->ws.release();
  delete h;
  return AMDR_OK;
}

int amdr_dense_small_approx_device(amdr_dense_small_t* h, const float* Q_dev, int32_t nq, float* S_dev, int64_t ldS,
                                   float* eps_dev, void* stream) {
  AMDR_REQUIRE(h != nullptr, "dense_small_approx: null handle");
  AMDR_REQUIRE(nq >= 0, "dense_small_approx: nq=%d", nq);
  if (nq == 0) return AMDR_OK;
  AMDR_REQUIRE(Q_dev && S_dev, "dense_small_approx: null buffer");
  AMDR_REQUIRE(ldS >= h->n_pad && ldS % 4 == 0, "dense_small_approx: ldS=%lld must be a multiple of 4 and >= %lld (rows padded to 32)",
               (long long)ldS, (long long)h->n_pad);
  AMDR_REQUIRE(h->ok, "dense_small_approx: the chunk matrix holds non-finite values or is out of the fp16 scale range");
  std::lock_guard<std::mutex> g(h->mu);
  AMDR_HIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const int ks = h->d / 128, n_tiles = (int)(h->n_pad / 32), n_qtiles = (nq + 31) / 32;
  size_t qh_bytes = 0;
  int rc = h->ws.ensure(dsh_ws_bytes(h->d, nq, &qh_bytes));
  if (rc) return rc;
  unsigned char* Qh = h->ws.as<unsigned char>();
  float* q_unscale = reinterpret_cast<float*>(Qh + qh_bytes);
  hipLaunchKernelGGL(dsh_split_queries_kernel, dim3(ceil_div(nq, 32)), dim3(256), 0, st, Q_dev, nq, h->d, Qh, q_unscale,
                     h->row_norm_max * h->x_scale, 1.f / h->x_scale, eps_dev);
  constexpr int NBUF = 4;
  AMDR_HIP(hipFuncSetAttribute((const void*)dsh_scores_kernel<NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * kDsStage));
  const int qgroups8 = ceil_div(ceil_div(n_qtiles, 2 * kDsWaves), 8) * 8;
  hipLaunchKernelGGL((dsh_scores_kernel<NBUF>), dim3(qgroups8 * ceil_div(n_tiles, kDsG)), dim3(kDsWaves * 64),
                     NBUF * kDsStage, st, h->img.as<unsigned char>(), (long)h->n_pad, n_tiles, ks, Qh, nq, q_unscale,
                     1.f / h->x_scale, S_dev, (long)ldS);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

}  // extern "C"
