// Wave-level exact top-k selection for gfx950 (64-wide wavefronts).
//
// Each wave owns a staging buffer of `cap` candidates in LDS (cap = power of
// two >= k + 64).  Candidates better than the wave's current threshold are
// appended with a ballot/prefix-popcount; when the buffer would overflow the
// wave sorts it in place (bitonic, descending), keeps the best k and raises
// the threshold to the k-th best.  No block barrier is involved, so the waves
// of a scan kernel never wait on each other inside the streaming loop.
// Selection is exact for any input order (the data only changes how often a
// wave compacts) and deterministic: candidates are totally ordered.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace amdr {

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- order-preserving float keys ----------------------------------------
__device__ __forceinline__ u32 ord32(float x) {
  x = x + 0.0f;  // -0.0 -> +0.0 so that equal scores compare equal
  u32 u = __float_as_uint(x);
  if (x != x) return 1u;  // NaN sorts last (but above padding)
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord32(u32 k) {
  u32 u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ u64 ord64(double x) {
  x = x + 0.0;
  u64 u = (u64)__double_as_longlong(x);
  if (x != x) return 1ull;
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unord64(u64 k) {
  u64 u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

// ---- candidate types ------------------------------------------------------
// C32: fp32 score + 32-bit id packed in one u64: (ord32(score) << 32) | ~id.
// Larger composite == better (higher score, then lower id).  0 == padding.
struct C32 {
  u64 c;
  __device__ __forceinline__ static C32 make(float s, u32 id) {
    C32 r;
    r.c = ((u64)ord32(s) << 32) | (u64)(0xffffffffu - id);
    return r;
  }
  __device__ __forceinline__ static C32 pad() {
    C32 r;
    r.c = 0ull;
    return r;
  }
  __device__ __forceinline__ bool is_pad() const { return c == 0ull; }
  __device__ __forceinline__ float score() const { return unord32((u32)(c >> 32)); }
  __device__ __forceinline__ long long id() const { return (long long)(0xffffffffu - (u32)c); }
};
__device__ __forceinline__ bool better(const C32& a, const C32& b) { return a.c > b.c; }

// C64: 64-bit key (ord64 of an fp64 score, or ord32 of an fp32 one) + 64-bit id.
struct C64 {
  u64 key;
  long long idv;
  __device__ __forceinline__ static C64 make(double s, long long id) {
    C64 r;
    r.key = ord64(s);
    r.idv = id;
    return r;
  }
  __device__ __forceinline__ static C64 make32(float s, long long id) {
    C64 r;
    r.key = (u64)ord32(s);
    r.idv = id;
    return r;
  }
  __device__ __forceinline__ static C64 pad() {
    C64 r;
    r.key = 0ull;
    r.idv = 0x7fffffffffffffffll;
    return r;
  }
  __device__ __forceinline__ bool is_pad() const { return key == 0ull; }
};
__device__ __forceinline__ bool better(const C64& a, const C64& b) {
  return a.key > b.key || (a.key == b.key && a.idv < b.idv);
}

// LDS traffic between lanes of ONE wave: the LDS unit serves a wave's
// operations in order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Sorts buf[0 .. cap) descending; cap must be a power of two (callers pass the smallest
// power of two covering the staged candidates, not the whole buffer).
template <class C>
__device__ inline void wave_bitonic_sort_desc(C* buf, int cap, int lane) {
  for (int size = 2; size <= cap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      wave_lds_fence();
      for (int t = lane; t < (cap >> 1); t += 64) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);
        C a = buf[lo], b = buf[hi];
        bool swap = desc ? better(b, a) : better(a, b);
        if (swap) {
          buf[lo] = b;
          buf[hi] = a;
        }
      }
    }
  }
  wave_lds_fence();
}

template <class C>
struct WaveTopK {
  C* buf;   // LDS, cap entries, private to this wave
  int cap;  // power of two >= k + 64
  int k;
  int cnt;  // wave-uniform
  C thr;    // wave-uniform; candidates must beat it to be staged

  __device__ __forceinline__ void init(C* b, int cap_, int k_) {
    buf = b;
    cap = cap_;
    k = k_;
    cnt = 0;
    thr = C::pad();
  }

  __device__ inline void compact(int lane) {
    // sort only the smallest power of two that covers the staged candidates
    int m = 2;
    while (m < cnt) m <<= 1;
    for (int i = cnt + lane; i < m; i += 64) buf[i] = C::pad();
    wave_bitonic_sort_desc(buf, m, lane);
    if (cnt > k) cnt = k;
    if (cnt == k) thr = buf[k - 1];
  }

  // every lane may offer one candidate
  __device__ __forceinline__ void push_lanes(C c, bool valid, int lane) {
    bool pass = valid && better(c, thr);
    u64 m = __ballot(pass);
    if (m == 0ull) return;
    int tot = __popcll(m);
    if (cnt + tot > cap) {
      compact(lane);
      pass = valid && better(c, thr);
      m = __ballot(pass);
      if (m == 0ull) return;
      tot = __popcll(m);
    }
    u64 lt = (lane == 0) ? 0ull : (m & (~0ull >> (64 - lane)));
    if (pass) buf[cnt + __popcll(lt)] = c;
    cnt += tot;
  }

  // one wave-uniform candidate (value identical in all lanes)
  __device__ __forceinline__ void push_uniform(C c, int lane) {
    if (better(c, thr)) {
      if (cnt == cap) compact(lane);
      if (better(c, thr)) {
        if (lane == 0) buf[cnt] = c;
        cnt += 1;
      }
    }
  }

  // after this buf[0 .. cnt) is the sorted (best first) top-min(k, seen)
  __device__ inline void finalize(int lane) { compact(lane); }
};

template <int WAVES>
__device__ __forceinline__ void block_sync() {
  if (WAVES == 1)
    wave_lds_fence();
  else
    __syncthreads();
}

// ---------------------------------------------------------------------------
// Short rows (n <= 64 * V, k <= 64): selection entirely in registers.
//   1. every lane holds V candidates and takes its local best;
//   2. the 64 local bests are sorted across the lanes (bitonic network on shuffles);
//      the k-th of them, T, is a lower bound of the k-th best overall (k candidates >= T
//      exist), so the answer lies among the candidates that are not worse than T — in
//      expectation only a few more than k;
//   3. those are compacted (ballot + prefix popcount) into a 64-entry LDS scratch, sorted
//      with the same network, and the first k are the result.
// Returns the number of valid results written, or -1 when more than 64 candidates survive
// step 2 (caller falls back to the staged selector).  ~3x fewer instructions than two
// LDS bitonic sorts for n ~ 600, k = 10.
__device__ __forceinline__ C32 wave_xchg(const C32& v, int partner) {
  C32 o;
  const unsigned lo = __shfl((unsigned)(v.c & 0xffffffffull), partner);
  const unsigned hi = __shfl((unsigned)(v.c >> 32), partner);
  o.c = ((u64)hi << 32) | lo;
  return o;
}
__device__ __forceinline__ C64 wave_xchg(const C64& v, int partner) {
  C64 o;
  const unsigned a = __shfl((unsigned)(v.key & 0xffffffffull), partner);
  const unsigned b = __shfl((unsigned)(v.key >> 32), partner);
  const unsigned c = __shfl((unsigned)((u64)v.idv & 0xffffffffull), partner);
  const unsigned d = __shfl((unsigned)((u64)v.idv >> 32), partner);
  o.key = ((u64)b << 32) | a;
  o.idv = (long long)(((u64)d << 32) | c);
  return o;
}

template <class C>
__device__ inline C wave_sort64_desc(C v, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const C o = wave_xchg(v, lane ^ stride);
      const bool keep_better = (((lane & size) == 0) == ((lane & stride) == 0));
      const bool mine_better = better(v, o);
      v = (keep_better == mine_better) ? v : o;
    }
  }
  return v;
}

template <class C, int V>
__device__ inline int wave_select_small(const C (&keys)[V], int k, C* scratch /* LDS, 64 entries */, int lane,
                                        C* out_sorted /* LDS or registers' spill target: k entries */) {
  C lbest = C::pad();
#pragma unroll
  for (int v = 0; v < V; ++v)
    if (better(keys[v], lbest)) lbest = keys[v];
  const C sorted_best = wave_sort64_desc(lbest, lane);
  const C T = wave_xchg(sorted_best, k - 1 < 63 ? k - 1 : 63);  // k-th local best (pad if fewer)
  int cnt = 0;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const bool pass = !keys[v].is_pad() && !better(T, keys[v]);
    const u64 m = __ballot(pass);
    const int tot = __popcll(m);
    if (cnt + tot > 64) return -1;
    const u64 lt = (lane == 0) ? 0ull : (m & (~0ull >> (64 - lane)));
    if (pass) scratch[cnt + __popcll(lt)] = keys[v];
    cnt += tot;
  }
  wave_lds_fence();
  C c = (lane < cnt) ? scratch[lane] : C::pad();
  c = wave_sort64_desc(c, lane);
  wave_lds_fence();
  if (lane < k) out_sorted[lane] = c;
  wave_lds_fence();
  return cnt < k ? cnt : k;
}

// Merge the finalized lists of all waves of a block into wave 0's list.
// Call with all threads; contains block barriers.  lists: [nwaves][stride]
// where each wave's WaveTopK.buf == lists + wave*stride.
template <class C>
__device__ inline void block_combine_topk(WaveTopK<C>& tk, C* lists, int stride, int nwaves, int wave, int lane,
                                          int* cnts /* LDS int[nwaves] */) {
  if (lane == 0) cnts[wave] = tk.cnt;
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < nwaves; ++w) {
      int n = cnts[w];
      for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool v = i < n;
        C c = v ? lists[w * stride + i] : C::pad();
        tk.push_lanes(c, v, lane);
      }
    }
    tk.finalize(lane);
  }
  __syncthreads();
}

}  // namespace amdr
