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

// Value of lane (l ^ S) for a compile-time S, without the LDS crossbar: ds_bpermute (what
// __shfl compiles to) costs a ~100-cycle LDS round trip per 32-bit word, and the selection
// networks below are chains of dependent exchanges.  Strides inside a 16-lane row are DPP
// moves (1-2 VALU ops); strides 16 and 32 use gfx950's v_permlane{16,32}_swap, which leaves
// {own, partner} in its two results in a lane-dependent order — x ^ a ^ b is the partner either way.
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
template <int S>
__device__ __forceinline__ u32 lane_xor(u32 x) {
  static_assert(S == 1 || S == 2 || S == 4 || S == 8 || S == 16 || S == 32, "stride");
  if constexpr (S == 1) {
    return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
  } else if constexpr (S == 2) {
    return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
  } else if constexpr (S == 4) {
    const int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xf, 0x5, false);  // row_shl:4 -> banks 0,2
    return (u32)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xf, 0xa, false);          // row_shr:4 -> banks 1,3
  } else if constexpr (S == 8) {
    return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xf, 0xf, false);  // row_ror:8
  } else if constexpr (S == 16) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return x ^ r.x ^ r.y;
  } else {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return x ^ r.x ^ r.y;
  }
}
// stride known after unrolling: the switch folds away
__device__ __forceinline__ u32 lane_xor_sw(u32 x, int s) {
  switch (s) {
    case 1: return lane_xor<1>(x);
    case 2: return lane_xor<2>(x);
    case 4: return lane_xor<4>(x);
    case 8: return lane_xor<8>(x);
    case 16: return lane_xor<16>(x);
    default: return lane_xor<32>(x);
  }
}
__device__ __forceinline__ u64 lane_xor_sw(u64 x, int s) {
  return ((u64)lane_xor_sw((u32)(x >> 32), s) << 32) | lane_xor_sw((u32)x, s);
}
__device__ __forceinline__ C32 wave_xchg_xor(const C32& v, int s) {
  C32 o;
  o.c = lane_xor_sw(v.c, s);
  return o;
}
__device__ __forceinline__ C64 wave_xchg_xor(const C64& v, int s) {
  C64 o;
  o.key = lane_xor_sw(v.key, s);
  o.idv = (long long)lane_xor_sw((u64)v.idv, s);
  return o;
}
// broadcast of one lane's value; `src` must be wave-uniform (v_readlane)
__device__ __forceinline__ u64 wave_bcast_u64(u64 x, int src) {
  const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)x, src);
  const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), src);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ C32 wave_bcast(const C32& v, int src) {
  C32 o;
  o.c = wave_bcast_u64(v.c, src);
  return o;
}
__device__ __forceinline__ C64 wave_bcast(const C64& v, int src) {
  C64 o;
  o.key = wave_bcast_u64(v.key, src);
  o.idv = (long long)wave_bcast_u64((u64)v.idv, src);
  return o;
}
// fp64 max / min as ONE instruction.  fmax()/fmin() make hipcc canonicalise both operands
// first (a v_max_f64 x, x, x each) because it cannot prove them free of signalling NaNs across
// a loop; every value that reaches these helpers was produced by arithmetic, so it is.
__device__ __forceinline__ double max_f64_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double min_f64_raw(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {  // every lane valid for the controls used below
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// All-lanes max / min of an fp64: four DPP steps reduce each 16-lane row (quad pairs, quad
// halves, then row_half_mirror and row_mirror — any pairing that joins the halves will do for
// a reduction), the four row results are read into scalar registers and combined.  The
// result is wave-uniform by construction (the compiler keeps it in SGPRs).
#define AMDR_WAVE_REDUCE_F64(NAME, OP)                                                        \
  __device__ __forceinline__ double NAME(double x) {                                          \
    x = OP(x, dpp_f64<0xB1>(x));  /* quad_perm [1,0,3,2] */                                   \
    x = OP(x, dpp_f64<0x4E>(x));  /* quad_perm [2,3,0,1] */                                   \
    x = OP(x, dpp_f64<0x141>(x)); /* row_half_mirror */                                       \
    x = OP(x, dpp_f64<0x140>(x)); /* row_mirror */                                            \
    const int lo = __double2loint(x), hi = __double2hiint(x);                                 \
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));   \
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16)); \
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)); \
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48)); \
    return OP(OP(r0, r1), OP(r2, r3));                                                        \
  }
AMDR_WAVE_REDUCE_F64(wave_allmax_f64, max_f64_raw)
AMDR_WAVE_REDUCE_F64(wave_allmin_f64, min_f64_raw)
#undef AMDR_WAVE_REDUCE_F64
// The same reductions inside aligned groups of W lanes (W a power of two <= 64): butterfly over
// the strides below W only.
template <int W>
__device__ __forceinline__ double seg_allmax_f64(double x) {
#pragma unroll
  for (int s = 1; s < W; s <<= 1)
    x = max_f64_raw(x, __longlong_as_double((long long)lane_xor_sw((u64)__double_as_longlong(x), s)));
  return x;
}
template <int W>
__device__ __forceinline__ double seg_allmin_f64(double x) {
#pragma unroll
  for (int s = 1; s < W; s <<= 1)
    x = min_f64_raw(x, __longlong_as_double((long long)lane_xor_sw((u64)__double_as_longlong(x), s)));
  return x;
}
template <int W>
__device__ __forceinline__ int seg_allsum_i32(int x) {
#pragma unroll
  for (int s = 1; s < W; s <<= 1) x += (int)lane_xor_sw((u32)x, s);
  return x;
}
__device__ __forceinline__ int wave_allsum_i32(int x) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) x += (int)lane_xor_sw((u32)x, s);
  return x;
}
__device__ __forceinline__ int wave_allmin_i32(int x) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int o = (int)lane_xor_sw((u32)x, s);
    x = o < x ? o : x;
  }
  return x;
}

// K32: a bare 32-bit ordered score key (ord32; 0 = no candidate) for the steps of a selection that do not need
// the id yet: one register, one exchange and one compare per step instead of two / two / three.
struct K32 {
  u32 c;
};
__device__ __forceinline__ bool better(const K32& a, const K32& b) { return a.c > b.c; }
__device__ __forceinline__ K32 wave_xchg_xor(const K32& v, int s) {
  K32 o;
  o.c = lane_xor_sw(v.c, s);
  return o;
}

// Bitonic network across the lanes: sorts every aligned group of N lanes (N = 16, 32 or 64)
// descending; exchange steps never reach past the group.  (The last merge runs in the same
// direction in EVERY group: with the textbook alternation the odd groups would come out ascending.)
template <class C, int N>
__device__ inline C wave_sortN_desc(C v, int lane) {
#pragma unroll
  for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const C o = wave_xchg_xor(v, stride);
      const bool keep_better = (((size == N) || (lane & size) == 0) == ((lane & stride) == 0));
      const bool mine_better = better(v, o);
      v = (keep_better == mine_better) ? v : o;
    }
  }
  return v;
}
template <class C>
__device__ inline C wave_sort64_desc(C v, int lane) {
  return wave_sortN_desc<C, 64>(v, lane);
}

template <class C, int V>
__device__ inline int wave_select_small(const C (&keys)[V], int k, C* scratch /* LDS, 64 entries */, int lane,
                                        C* out_sorted /* LDS or registers' spill target: k entries */) {
  C lbest = C::pad();
#pragma unroll
  for (int v = 0; v < V; ++v)
    if (better(keys[v], lbest)) lbest = keys[v];
  const C sorted_best = wave_sort64_desc(lbest, lane);
  const C T = wave_bcast(sorted_best, k - 1 < 63 ? k - 1 : 63);  // k-th local best (pad if fewer)
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
  // the survivors sit in lanes 0 .. cnt-1 (a few more than k): sort only the lane group that holds them
  if (cnt <= 16)
    c = wave_sortN_desc<C, 16>(c, lane);
  else if (cnt <= 32)
    c = wave_sortN_desc<C, 32>(c, lane);
  else
    c = wave_sortN_desc<C, 64>(c, lane);
  wave_lds_fence();
  if (lane < k) out_sorted[lane] = c;
  wave_lds_fence();
  return cnt < k ? cnt : k;
}

// The same selector for TWO independent key sets per wave: lanes 0-31 rank one, lanes 32-63 the
// other (every cross-lane step stays inside its 32-lane half).  One instruction stream then
// serves two queries, and the lane-best sort has 15 exchange stages instead of 21: measured on
// the UCC-en dense top-k (591 keys per query, k = 10) the per-query cost fell by about 40 %.
// k <= 32.  Returns the survivors of THIS lane's half (min(cnt, k) of them are valid results, best
// first, in lanes 0..k-1 of the half: `out`), or -1 for the whole wave if either half had more
// than 32 survivors (mass ties at the cut) — the caller then ranks both by the general path.
// Implementation on 32-bit score keys (the first version carried 64-bit (score, ~id) keys through every step:
// 29.0 -> 27.6 us per 37 376 UCC-en queries): the lane best, the sort of the 32 lane bests and the cut look at
// the score only (v_max_u32 / one exchange + one compare per stage); the 64-bit (score, ~id) candidates are built
// for the survivors alone — every key whose SCORE reaches the cut takes part, so equal scores at the cut are all
// there and the final 64-bit sort puts the lower id first, exactly as the 64-bit selector does.
// sk[v]: ord32 of the score, 0 for "no candidate"; id_of(v): the candidate's id.
template <int V, class IdOf>
__device__ inline int wave_select_small_pair32(const u32 (&sk)[V], IdOf&& id_of, int k, C32* scratch /* LDS, 64 */,
                                               int lane, C32& out) {
  K32 lb;
  lb.c = 0u;
#pragma unroll
  for (int v = 0; v < V; ++v) lb.c = sk[v] > lb.c ? sk[v] : lb.c;
  const K32 sorted_best = wave_sortN_desc<K32, 32>(lb, lane);
  const int src = (lane & 32) + (k - 1 < 31 ? k - 1 : 31);  // k-th lane best of this half
  const u32 T = (u32)__shfl((int)sorted_best.c, src);
  // Survivors -> scratch: every lane counts its own (one compare + add per key), an inclusive prefix sum over the
  // 32 lanes of the half gives each lane its first slot, then the lane stores its survivors one after the other
  // (their order in the scratch does not matter: they are sorted next).  The first version ran a ballot, two
  // popcounts, two mbcnt and an overflow vote PER KEY REGISTER (~12 vector instructions x V against ~6 here; the
  // kernel is exactly vector-issue-bound).
  const u32 Te = T > 1u ? T : 1u;  // "not a candidate" is key 0
  int mine = 0;
#pragma unroll
  for (int v = 0; v < V; ++v) mine += (sk[v] >= Te) ? 1 : 0;
  int incl = mine;
  const int sl = lane & 31;
#pragma unroll
  for (int s = 1; s < 32; s <<= 1) {
    const int o = __shfl_up(incl, s, 32);
    incl += (sl >= s) ? o : 0;
  }
  const int cnt = __shfl(incl, (lane & 32) + 31);  // survivors of this lane's half
  if (__ballot(cnt > 32)) return -1;
  int at = (lane & 32) + incl - mine;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    if (sk[v] >= Te) {
      C32 c;
      c.c = ((u64)sk[v] << 32) | (u64)(0xffffffffu - (u32)id_of(v));
      scratch[at++] = c;
    }
  }
  wave_lds_fence();
  C32 c = ((lane & 31) < cnt) ? scratch[lane] : C32::pad();
  if (__ballot((lane & 31) >= 16 && (lane & 31) < cnt) == 0ull)
    c = wave_sortN_desc<C32, 16>(c, lane);
  else
    c = wave_sortN_desc<C32, 32>(c, lane);
  wave_lds_fence();
  out = c;
  return cnt;
}

// Top-k of a row of <= 64 V fp32 scores by ONE wave in registers (wave_select_small): keys (score, column), the sorted
// list in buf[0 .. k), scratch buf[64 .. 128).  -1: mass ties at the cut, the caller takes the staged selector.
constexpr int kSelectRowsMax = 2048;  // rows one wave ranks in registers (<= 32 keys per lane)
template <int V>
__device__ __forceinline__ int select_row(const float* __restrict__ row, long lo, long hi, int k, int lane, C32* buf) {
  C32 keys[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const long r = lo + lane + 64 * v;
    keys[v] = (r < hi) ? C32::make(row[r], (u32)r) : C32::pad();
  }
  return wave_select_small<C32, V>(keys, k, buf + 64, lane, buf);
}

typedef float tk_v4f __attribute__((ext_vector_type(4)));

// Two queries per wave (lanes 0-31 / 32-63), for rows of <= 1024 scores at k <= 32 when there is a
// single slab — the serving corpora under a long batch.  grid.x = ceil(nq / 2); a wave whose
// selector reports mass ties (-1) ranks its two rows one after the other with the staged selector.
template <int V>
__device__ __forceinline__ int select_row_pair(const float* __restrict__ S, long ldS, long n, int q, bool has_q, int k,
                                               int lane, C32* scratch, C32& out) {
  static_assert(V % 4 == 0, "four consecutive scores per 16-byte load");
  const float* row = S + (size_t)q * ldS;
  const int j = lane & 31;
  // 16-byte loads (a half-wave covers 512 B of its row per instruction): the kernel waits on memory
  // for two thirds of its life, so fewer, wider requests in flight earlier is what shortens it.
  // Rows are padded to ldS (a multiple of 32 floats), so a whole float4 below ldS is inside the row.
  tk_v4f blk[V / 4];
#pragma unroll
  for (int u = 0; u < V / 4; ++u) {
    const long c0 = 128L * u + 4 * j;
    const tk_v4f z = {0.f, 0.f, 0.f, 0.f};
    blk[u] = (has_q && c0 < ldS) ? __builtin_nontemporal_load(reinterpret_cast<const tk_v4f*>(row + c0)) : z;  // read once
  }
  u32 sk[V];  // score keys only; ids (128 u + 4 j + e) are attached to the survivors (wave_select_small_pair32)
#pragma unroll
  for (int u = 0; u < V / 4; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long r = 128L * u + 4 * j + e;
      sk[4 * u + e] = (has_q && r < n) ? ord32(blk[u][e]) : 0u;
    }
  return wave_select_small_pair32<V>(sk, [&](int v) { return 128 * (v >> 2) + 4 * j + (v & 3); }, k, scratch, lane, out);
}

// rows of <= 256 / 512 / 640 / 1024 scores: 8 / 16 / 20 / 32 keys per lane
__device__ __forceinline__ int select_row_pair_any(const float* __restrict__ S, long ldS, long n, int q, bool has_q,
                                                   int k, int lane, C32* scratch, C32& out) {
  if (n <= 256) return select_row_pair<8>(S, ldS, n, q, has_q, k, lane, scratch, out);
  if (n <= 512) return select_row_pair<16>(S, ldS, n, q, has_q, k, lane, scratch, out);
  if (n <= 640) return select_row_pair<20>(S, ldS, n, q, has_q, k, lane, scratch, out);
  return select_row_pair<32>(S, ldS, n, q, has_q, k, lane, scratch, out);
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
