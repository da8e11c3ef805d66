// Row-sharded corpus, the two kernels either side of the RCCL all-gather (the reference is
// single-process: legalrag/config.py:106 `colbert_nranks = 1` is its only mention of ranks).
//
//   shard_pack_kernel   every channel's (score, LOCAL id) lists of this rank -> ONE int64 send buffer
//                       row q = [ch0: k0 score words | k0 global ids | ch1: ... ]   (score word = the bits of the score
//                       as fp64 — an fp32 score widens exactly —, global id = local id + offset, -1 stays -1)
//   shard_merge_kernel  reads the gathered [world][nq][row] buffer IN PLACE: one wave per (query, channel) ranks the
//                       world * k candidates by (score desc, global id asc) and writes the channel's global top-k.
//
// One launch each, whatever the number of channels: the torch form of the same exchange was ~20 small launches (where /
// to / cat / contiguous per channel, then one merge launch per channel).  The merge is latency-bound (80 candidates per
// wave at world = 8, k = 10): candidates live in registers, one or two per lane, and are ranked by a bitonic network on
// DPP / permlane exchanges; longer lists take the staged selector of topk.hpp.
#include "common.hpp"
#include "topk.hpp"

#include <cfloat>

namespace amdr {

constexpr int kShardChans = 4;
struct ShardLayout {
  const void* scores[kShardChans];      // pack: this rank's lists
  const long long* ids[kShardChans];
  void* out_scores[kShardChans];        // merge: the global lists
  long long* out_ids[kShardChans];
  int k[kShardChans], f64[kShardChans], col[kShardChans];
  int n, row;                           // channels, int64 words per query row
};

// element `c` of a 4-entry kernel-argument array by a compare chain: a dynamic index would copy the whole argument struct
// to scratch memory in every thread (measured: the k = 10 merge of 7 128 waves took 40 us with it)
#define SHARD_PICK(ARR, c) ((c) == 0 ? (ARR)[0] : (c) == 1 ? (ARR)[1] : (c) == 2 ? (ARR)[2] : (ARR)[3])

__global__ __launch_bounds__(256) void shard_pack_kernel(ShardLayout L, int nq, long long offset,
                                                         long long* __restrict__ send) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)nq * L.row) return;
  const int q = (int)(idx / L.row), j = (int)(idx - (long)q * L.row);
  int ch = 0;
#pragma unroll
  for (int c = 1; c < kShardChans; ++c)
    if (c < L.n && j >= L.col[c]) ch = c;
  const int k = SHARD_PICK(L.k, ch), jj = j - SHARD_PICK(L.col, ch);
  const void* sp = SHARD_PICK(L.scores, ch);
  const long long* ip = SHARD_PICK(L.ids, ch);
  long long w;
  if (jj < k) {
    const double s = SHARD_PICK(L.f64, ch) ? reinterpret_cast<const double*>(sp)[(size_t)q * k + jj]
                                           : (double)reinterpret_cast<const float*>(sp)[(size_t)q * k + jj];
    w = __double_as_longlong(s);
  } else {
    const long long id = ip[(size_t)q * k + (jj - k)];
    w = id >= 0 ? id + offset : id;
  }
  send[idx] = w;
}

__device__ __forceinline__ C64 shard_cand(const long long* __restrict__ g, int nq, int row, int q, int col, int k, int i,
                                          int total) {
  if (i >= total) return C64::pad();
  const int p = i / k, j = i - p * k;
  const long long* r = g + ((size_t)p * nq + q) * row + col;
  const long long id = r[k + j];
  if (id < 0) return C64::pad();
  return C64::make(__longlong_as_double(r[j]), id);
}

// Bitonic network across the 64 lanes on (key, id) pairs kept as two separate registers pairs: selecting a 16-byte
// struct with `cond ? v : o` (topk.hpp wave_sortN_desc<C64>) made hipcc park both in scratch memory and load one back
// through a computed address — 40 B of scratch per lane and a memory round trip per stage.
__device__ __forceinline__ void cx64(u64& key, long long& idv, int stride, bool keep_better) {
  const u64 ok = lane_xor_sw(key, stride);
  const long long oi = (long long)lane_xor_sw((u64)idv, stride);
  const bool mine_better = key > ok || (key == ok && idv < oi);
  const bool keep = keep_better == mine_better;
  key = keep ? key : ok;
  idv = keep ? idv : oi;
}
__device__ __forceinline__ void shard_sort64_desc(u64& key, long long& idv, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1)
      cx64(key, idv, stride, ((size == 64) || (lane & size) == 0) == ((lane & stride) == 0));
  }
}
// sorted (descending) 64-lane sequence from a bitonic one: the merge half of the network
__device__ __forceinline__ void shard_bitonic_merge64_desc(u64& key, long long& idv, int lane) {
#pragma unroll
  for (int stride = 32; stride > 0; stride >>= 1) cx64(key, idv, stride, (lane & stride) == 0);
}

template <bool STAGED>
__global__ __launch_bounds__(256) void shard_merge_kernel(ShardLayout L, int world, int nq,
                                                          const long long* __restrict__ g, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long w = (long)blockIdx.x * 4 + wave;
  if (w >= (long)nq * L.n) return;
  const int q = (int)(w / L.n), ch = __builtin_amdgcn_readfirstlane((int)(w - (long)q * L.n));  // wave-uniform
  const int k = SHARD_PICK(L.k, ch), col = SHARD_PICK(L.col, ch), total = world * k;
  const bool f64 = SHARD_PICK(L.f64, ch) != 0;
  void* osp = SHARD_PICK(L.out_scores, ch);
  long long* oip = SHARD_PICK(L.out_ids, ch);
  u64 bkey = 0ull;  // lane j < k: the j-th hit
  long long bid = 0;
  if (!STAGED || (total <= 128 && k <= 64)) {
    const C64 a = shard_cand(g, nq, L.row, q, col, k, lane, total);
    bkey = a.key, bid = a.idv;
    shard_sort64_desc(bkey, bid, lane);
    if (total > 64) {
      const C64 b = shard_cand(g, nq, L.row, q, col, k, 64 + lane, total);
      u64 k2 = b.key;
      long long i2 = b.idv;
      shard_sort64_desc(k2, i2, lane);
      // the second half reversed (lane l <-> 63 - l = every stride flipped): max(a[l], b[63 - l]) is the top 64 of the
      // union as a bitonic sequence
#pragma unroll
      for (int st = 1; st < 64; st <<= 1) {
        k2 = lane_xor_sw(k2, st);
        i2 = (long long)lane_xor_sw((u64)i2, st);
      }
      const bool mine = bkey > k2 || (bkey == k2 && bid < i2);
      bkey = mine ? bkey : k2;
      bid = mine ? bid : i2;
      shard_bitonic_merge64_desc(bkey, bid, lane);
    }
  } else {
    C64* buf = reinterpret_cast<C64*>(smem) + (size_t)wave * cap;
    WaveTopK<C64> tk;
    tk.init(buf, cap, k);
    for (int base = 0; base < total; base += 64) {
      const C64 c = shard_cand(g, nq, L.row, q, col, k, base + lane, total);
      tk.push_lanes(c, !c.is_pad(), lane);
    }
    tk.finalize(lane);
    const int got = tk.cnt;
    for (int j = lane; j < k; j += 64) {
      const C64 c = j < got ? buf[j] : C64::pad();
      const bool v = !c.is_pad();
      if (f64)
        reinterpret_cast<double*>(osp)[(size_t)q * k + j] = v ? unord64(c.key) : -DBL_MAX;
      else
        reinterpret_cast<float*>(osp)[(size_t)q * k + j] = v ? (float)unord64(c.key) : -FLT_MAX;
      oip[(size_t)q * k + j] = v ? c.idv : -1ll;
    }
    return;
  }
  if (lane < k) {
    const bool v = bkey != 0ull;
    if (f64)
      reinterpret_cast<double*>(osp)[(size_t)q * k + lane] = v ? unord64(bkey) : -DBL_MAX;
    else
      reinterpret_cast<float*>(osp)[(size_t)q * k + lane] = v ? (float)unord64(bkey) : -FLT_MAX;
    oip[(size_t)q * k + lane] = v ? bid : -1ll;
  }
}

static int shard_layout(const amdr_shard_chan_t* chans, int n_chan, bool merge, ShardLayout* L) {
  AMDR_REQUIRE(chans != nullptr && n_chan >= 1 && n_chan <= kShardChans, "shard: %d channels outside [1,%d]", n_chan,
               kShardChans);
  int col = 0;
  for (int c = 0; c < kShardChans; ++c) {
    L->scores[c] = nullptr, L->ids[c] = nullptr, L->out_scores[c] = nullptr, L->out_ids[c] = nullptr;
    L->k[c] = 0, L->f64[c] = 0, L->col[c] = 0;
  }
  for (int c = 0; c < n_chan; ++c) {
    AMDR_REQUIRE(chans[c].k >= 1 && chans[c].k <= AMDR_MAX_K, "shard: channel %d depth %d outside [1,%d]", c, chans[c].k,
                 AMDR_MAX_K);
    AMDR_REQUIRE(chans[c].scores && chans[c].ids, "shard: channel %d has a null buffer", c);
    if (merge) {
      L->out_scores[c] = chans[c].scores;
      L->out_ids[c] = reinterpret_cast<long long*>(chans[c].ids);
    } else {
      L->scores[c] = chans[c].scores;
      L->ids[c] = reinterpret_cast<const long long*>(chans[c].ids);
    }
    L->k[c] = chans[c].k;
    L->f64[c] = chans[c].f64 ? 1 : 0;
    L->col[c] = col;
    col += 2 * chans[c].k;
  }
  L->n = n_chan;
  L->row = col;
  return AMDR_OK;
}

}  // namespace amdr

using namespace amdr;

extern "C" {

int amdr_shard_row_words(const amdr_shard_chan_t* chans, int32_t n_chan, int64_t* words) {
  AMDR_REQUIRE(words != nullptr, "shard_row_words: null");
  ShardLayout L;
  amdr_shard_chan_t tmp[kShardChans];
  AMDR_REQUIRE(chans != nullptr && n_chan >= 1 && n_chan <= kShardChans, "shard_row_words: bad channels");
  static int64_t dummy;  // the layout only needs the depths: any non-null pointers pass its checks
  for (int c = 0; c < n_chan; ++c) tmp[c] = amdr_shard_chan_t{&dummy, &dummy, chans[c].k, chans[c].f64};
  int rc = shard_layout(tmp, n_chan, false, &L);
  if (rc) return rc;
  *words = L.row;
  return AMDR_OK;
}

int amdr_shard_pack_device(const amdr_shard_chan_t* chans, int32_t n_chan, int32_t nq, int64_t id_offset, int64_t* send,
                           int32_t device, void* stream) {
  ShardLayout L;
  int rc = shard_layout(chans, n_chan, false, &L);
  if (rc) return rc;
  AMDR_REQUIRE(nq >= 0 && id_offset >= 0, "shard_pack: nq=%d offset=%lld", nq, (long long)id_offset);
  if (nq == 0) return AMDR_OK;
  AMDR_REQUIRE(send != nullptr, "shard_pack: null send buffer");
  AMDR_HIP(hipSetDevice(device));
  const long total = (long)nq * L.row;
  hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L, nq,
                     (long long)id_offset, (long long*)send);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

int amdr_shard_merge_device(const int64_t* gathered, int32_t world, int32_t nq, const amdr_shard_chan_t* out_chans,
                            int32_t n_chan, int32_t device, void* stream) {
  ShardLayout L;
  int rc = shard_layout(out_chans, n_chan, true, &L);
  if (rc) return rc;
  AMDR_REQUIRE(world >= 1 && nq >= 0, "shard_merge: world=%d nq=%d", world, nq);
  if (nq == 0) return AMDR_OK;
  AMDR_REQUIRE(gathered != nullptr, "shard_merge: null gathered buffer");
  AMDR_HIP(hipSetDevice(device));
  bool staged = false;
  int cap = 0;
  for (int c = 0; c < n_chan; ++c) {
    if ((long)world * L.k[c] > 128 || L.k[c] > 64) staged = true;
    if (topk_cap(L.k[c]) > cap) cap = topk_cap(L.k[c]);
  }
  const long waves = (long)nq * n_chan;
  const dim3 grid((unsigned)((waves + 3) / 4));
  if (staged)
    hipLaunchKernelGGL(shard_merge_kernel<true>, grid, dim3(256), (size_t)4 * cap * sizeof(C64), (hipStream_t)stream, L, world,
                       nq, (const long long*)gathered, cap);
  else
    hipLaunchKernelGGL(shard_merge_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, L, world, nq,
                       (const long long*)gathered, cap);
  AMDR_HIP(hipGetLastError());
  return AMDR_OK;
}

}  // extern "C"
