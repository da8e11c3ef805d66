// Shared host-side plumbing for libamdretrieval (error strings, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "../../include/amdretrieval.h"

namespace amdr {

std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define AMDR_HIP(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return ::amdr::fail(_e == hipErrorOutOfMemory ? AMDR_ENOMEM : AMDR_EHIP, "%s failed: %s (%s:%d)", \
                          #expr, hipGetErrorString(_e), __FILE__, __LINE__);                    \
  } while (0)

#define AMDR_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return ::amdr::fail(AMDR_EINVAL, __VA_ARGS__); \
  } while (0)

// Growable device buffer owned by a handle (never shrinks).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return AMDR_OK;
    if (p) {
      AMDR_HIP(hipFree(p));
      p = nullptr;
      cap = 0;
    }
    AMDR_HIP(hipMalloc(&p, bytes));
    cap = bytes;
    return AMDR_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// top-k staging capacity: power of two, >= k + 64 (one wave of appends always fits)
inline int topk_cap(int k) { return next_pow2(k + 64) < 128 ? 128 : next_pow2(k + 64); }

int check_device(int device);

// A chunk matrix larger than the 256 MiB Infinity Cache is streamed with the non-temporal policy
// (read once per scan; nothing to keep on-die).  AMDR_DENSE_NT=0|1 pins the choice.
inline bool dense_stream_nontemporal(long n, int d) {
  const char* e = getenv("AMDR_DENSE_NT");
  if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
  return (size_t)n * (size_t)d * sizeof(float) > ((size_t)256 << 20);
}

// ---- batched (32-query tile, fp32 MFMA) dense path: dense_mfma.hip -----------
struct DenseMfmaPlan {
  int q_tiles, grid_x, grid_y, slabs, cap, waves;
  long rows_per_block, rows_per_slab, ld;  // ld = leading dimension of S (n rounded up to 32 floats)
  size_t lds_scores, s_bytes, part_bytes;
};
bool dense_mfma_supported(int d);
void dense_mfma_plan(long n, int d, int nq, int k, DenseMfmaPlan* p);
// mode 0: every row's score; 1: per-tile maxima; 2: re-scoring of the tiles in tile_list (dense_mfma.hip)
// `gate` (nullable device int): the launch does nothing unless *gate != 0 — the exact first pass behind the fp16 one
// (dense_hi.hip) is enqueued unconditionally and decides on the device whether it runs
int dense_mfma_launch_scores(const DenseMfmaPlan& p, const float* X, long n, int d, const float* Q, int nq, float* S,
                             hipStream_t st, int mode = 0, const int* tile_list = nullptr,
                             const int* tile_count = nullptr, long n_real = 0, const int* gate = nullptr,
                             int list_stride = 0);
// two-level top-k helpers: sorted unique list of the candidate tiles (a bitmap in LDS up to kUniqueBitmapTilesMax tiles,
// any number of candidates; beyond, a one-wave sort of <= 8 192 candidates); column -> row id of the final hits
constexpr int kUniqueBitmapTilesMax = 1 << 20;
int dense_tiles_unique_launch(const int64_t* tile_ids, int n_in, long n_tiles, int* list, int* count, hipStream_t st,
                              const int* gate = nullptr);
int dense_tiles_sort_per_query_launch(const int64_t* tile_ids, int m, int kc, int* list, int* count, hipStream_t st);
int dense_tiles_remap_launch(int64_t* ids, int total, const int* list, const int* count, long n_real, hipStream_t st,
                             const int* gate = nullptr, int k = 1, int list_stride = 0);
int dense_mfma_launch_topk(const DenseMfmaPlan& p, const float* S, long n, int nq, int k, void* part,
                           float* fin_scores, int64_t* fin_ids, hipStream_t st, const int* gate = nullptr);

// ---- fp16 first pass of the two-level top-k on large matrices: dense_hi.hip ----
bool dense_hi_supported(int d);
int dense_hi_max_queries(int d);  // queries per scan: 64, 48 at d = 1 024
long dense_hi_sample_stride(long n);
long dense_hi_sample_items(long n);
size_t dense_hi_mt_bytes(long n);                       // maxima of the sample, [items][64]
size_t dense_hi_cand_entries(long n, int m, int kc);    // 8-byte entries of the flat candidate list
int dense_hi_launch_sample(const float* X, long n, int d, const float* Q, int nq, float* MT, hipStream_t st, float x_scale);
int dense_hi_launch_transpose(const float* MT, long items, int nq, long ldM, float* M, hipStream_t st);
int dense_hi_launch_emit(const float* X, long n, int d, const float* Q, int nq, const float* tau, int tau_stride, void* cand,
                         unsigned int* total, size_t cap, hipStream_t st, float x_scale);
size_t dense_hi_cand_part_bytes(int m, int kc);
int dense_hi_launch_cand_topk(const void* cand, const unsigned int* total, size_t cap, int m, int kc, void* part,
                              int* nparts, hipStream_t st);
int dense_hi_launch_check(const float* vals, int64_t* ids, int m, int kc1, int k, const float* Q, int d,
                          float row_norm_max, float x_scale, long n_tiles, const unsigned int* total, size_t cap, int* flag,
                          unsigned int* unresolved, hipStream_t st);
int dense_stats_launch(const float* X, long n, int d, unsigned int* out2, hipStream_t st);
// round 4: the tail of a large search in four launches + two gated ones (dense_hi.hip, dense_mfma.hip)
long dense_hi2_sample_stride(long n, int qtiles);
long dense_hi2_sample_items(long n, int qtiles);
size_t dense_hi2_qcap(long n, int qtiles, int kc);
int dense_hi2_launch_sample(const float* X, long n, int d, const float* Q, int nq, int qtiles, float* MT, hipStream_t st,
                            float x_scale);
int dense_hi2_launch_tau(const float* MT, long n, int d, int nq, int qtiles, int kc, float* tau, unsigned int* qcount,
                         int* flag, unsigned int* stats, hipStream_t st);
int dense_hi2_launch_emit(const float* X, long n, int d, const float* Q, int nq, const float* tau, void* qlist,
                          unsigned int* qcount, size_t qcap, hipStream_t st, float x_scale, int qtiles);
int dense_hi2_launch_select(const void* qlist, const unsigned int* qcount, size_t qcap, int m, int kc, int k, const float* Q,
                            int d, float row_norm_max, float x_scale, long n_tiles, int* list, int* count, int* unres,
                            int* flag, unsigned int* unresolved, hipStream_t st);
int dense_hi2_launch_exact_select(const float* M, long ldM, long n_tiles, int m, int k, int list_stride, int* list, int* count,
                                  const int* unres, const int* gate, hipStream_t st);
int dense_rescore_tiles_launch(const float* X, long n_real, int d, const float* Q, int m, const int* list, const int* count,
                               int list_stride, int max_tiles, long ldS, float* S, hipStream_t st);
int dense_final_topk_launch(const float* S, long ldS, const int* list, const int* count, int list_stride, int max_tiles,
                            long n_real, int m, int k, float* fin_scores, int64_t* fin_ids, hipStream_t st);


// ---- dense top-k + fusion in one call (fuse.hip; dense.hip amdr_dense_search_fuse_device) ----
struct FuseTail {  // the fusion that follows the dense channel: parameters, the BM25 lists, the outputs (device pointers)
  const amdr_fuse_params_t* p;
  const int64_t* dense_row2uid;
  const int64_t* bm25_ids;
  const double* bm25_scores;
  int kb;
  const int64_t* bm25_row2uid;
  int64_t* out_ids;
  double* out_vals;
  int32_t* out_mask;
  int32_t* out_count;
};
// first pass of the two-pass long-batch form (dense_small_hi.hip): the fp16 image and statistics of a short chunk matrix
int dense_small_create_from(int device, const float* X, int64_t n, int d, amdr_dense_small_t** out);
int dense_small_reserve(amdr_dense_small_t* h, int nq_max);
bool dense_small_usable(const amdr_dense_small_t* h);
// second pass of the two-pass long-batch form (fuse.hip dense_hi_select_fuse_kernel): candidates inside the proven margin
// of the approximate scores in S, exact dots, top-k (+ the fusion when t != nullptr)
int dense_hi_select_launch(const FuseTail* t, int q0, const float* S, long ldS, long n, int m, int kd, const float* X,
                           const float* Q, int d, const float* eps, float* fin_scores, int64_t* fin_ids,
                           unsigned int* fallbacks, hipStream_t st);
// one kernel ranks the rows of S and fuses (dense_select_fuse_kernel): single slab, <= 1 024 rows, kd + kb <= 32
bool dense_select_fuse_applies(long n, int slabs, int m, int kd, int kb);
// queries [q0, q0 + m) of the batch: S holds their score rows; writes their dense lists and their fused outputs
int dense_select_fuse_launch(const FuseTail& t, int q0, const float* S, long ldS, long n, int m, int kd, int cap,
                             float* fin_scores, int64_t* fin_ids, hipStream_t st);
// the plain fusion launch over finished dense lists (the unfused path of the same call)
int dense_fuse_plain_launch(const FuseTail& t, int q0, int m, int kd, const float* dense_scores, const int64_t* dense_ids,
                            hipStream_t st);

// ---- the one-launch serving step (fuse.hip hybrid_small_kernel): what it needs from the two handles --------------
struct DenseRaw {
  const float* X;
  long n;
  int d;
  float* S;   // [nq][ld] score scratch of the "_device" workspace
  long ld;
};
// ensures the score scratch for nq rows (dense.hip)
int dense_small_raw(amdr_dense_t* h, int nq, DenseRaw* out);
struct Bm25Raw {
  const long long* term_ptr;
  const int* post_doc;
  const double* post_w;
  const double* idf;
  long n_terms, n_docs;
  int* ticket;   // [64] zeroed arrival counters of the one-launch step (self-resetting)
  int slab, nslabs, cap, nvt;
  size_t lds;
  bool argmax, select_on;
};
int bm25_small_raw(amdr_bm25_t* h, int nq, int k, Bm25Raw* out);  // (bm25.hip)
std::mutex& dense_mutex(amdr_dense_t* h);
std::mutex& bm25_mutex(amdr_bm25_t* h);
int dense_device_of(const amdr_dense_t* h);

// ---- long-batch dense path: dense_panel.hip (panel of chunk rows shared by a block through LDS) ----
struct DensePanelPlan {
  int parts, base, rem, nb, m_tiles, gm, waves;
  size_t lds;
};
bool dense_panel_supported(long n, int d, int nq);
void dense_panel_plan(long n, int d, int nq, DensePanelPlan* p);
int dense_panel_launch_scores(const DensePanelPlan& p, const float* X, long n, int d, const float* Q, int nq, long ldS,
                              float* S, hipStream_t st);

}  // namespace amdr
