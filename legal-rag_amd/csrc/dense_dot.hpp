// The GEMV form of a dense score: one wave per (query, row), lane l holds the float4 pieces at columns 4 l + 256 c, the
// 64 partial sums folded by six DPP row operations.  Shared by dense.hip (dense_scan_topk_kernel, dense_all_scores_kernel,
// dense_score_rows_kernel) and the one-launch serving step of fuse.hip: the same instruction sequence, hence the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace amdr {

// --- wave64 sum via DPP: result valid in lane 63 -----------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row = row sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = total
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}

// <Q[q], X[r]> as dense_all_scores_kernel takes it (d % 4 == 0); valid in lane 63
__device__ __forceinline__ float dense_row_dot(const float* __restrict__ xr, const float* __restrict__ qr, int d, int lane) {
  float acc = 0.f;
  for (int col = lane * 4; col < d; col += 256) {
    const float4 a = *reinterpret_cast<const float4*>(xr + col);
    const float4 b = *reinterpret_cast<const float4*>(qr + col);
    acc = dot4(a, b, acc);
  }
  return wave_sum_to_lane63(acc);
}

}  // namespace amdr
