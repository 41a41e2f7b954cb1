// HBM-bound helpers of the IMPALA CNN on the channel-blocked bf16 layout [frame][C/32][H][W][32] (gfx950).
//
//  vpt_pool_kernel   : F.max_pool2d(x, 3, stride 2, pad 1)  (lib/impala_cnn.py:117) for stacks 1..2, whose
//                      input is post-ReLU (>= 0), so the -inf padding can be 0 and the max is taken on the
//                      bf16 bit patterns as packed unsigned 16-bit integers.  Emits sum / sum-of-squares of
//                      the pooled frame for the following GroupNorm `n`.
//  vpt_affine_kernel : y = (x - mean_f) * rstd_f * gain + bias with whole-frame statistics, i.e.
//                      CnnDownStack.n = GroupNorm(1, C) (lib/impala_cnn.py:99-100,118-119; per-channel gain)
//                      and ImpalaCNN.dense's LayerNorm over the flattened C*H*W vector (lib/impala_cnn.py:177-184,
//                      lib/util.py:61-62; per-element gain, permuted host-side into the blocked order).
//                      Emits the statistics of y for the next GroupNorm.
// Every lane moves 16 bytes (8 bf16) per access, EW_ITEMS accesses per thread (all loads issued before the
// first use); one block-level reduction and one pair of fp64 atomics per block keeps the per-frame
// statistics off the critical path.
#include "vpt_common.h"
#include "vpt_kernels.h"

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#define EW_ITEMS 4
#define EW_PER_BLOCK (256 * EW_ITEMS)

__global__ __launch_bounds__(256) void vpt_pool_kernel(VptPoolArgs a) {
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int per_frame = a.CB * PH * PW * 4;  // 16-byte items
  const int blocks_per_frame = (per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK;
  const int f = blockIdx.x / blocks_per_frame;
  const int base = (blockIdx.x - f * blocks_per_frame) * EW_PER_BLOCK + threadIdx.x;
  float s_sum = 0.f, s_sq = 0.f;
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = base + it * 256;
    if (item >= per_frame) break;
    const int oct = item & 3;
    int r = item >> 2;
    const int px = r % PW; r /= PW;
    const int py = r % PH;
    const int cb = r / PH;
    const vpt_op16* plane = a.x + ((size_t)(f * a.CB + cb) * a.H * a.W) * 32 + oct * 8;
    u16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};
    u16x8 am = {15, 15, 15, 15, 15, 15, 15, 15};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = 2 * py + dy;
      if (y < 0 || y >= a.H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int x = 2 * px + dx;
        if (x < 0 || x >= a.W) continue;
        const u16x8 v = *(const u16x8*)(plane + (size_t)(y * a.W + x) * 32);
        if (a.argmax) {  // strict > keeps the FIRST maximum in scan order (torch's rule); an all-zero window keeps 15
          const unsigned short code = (unsigned short)((dy + 1) * 3 + (dx + 1));
#pragma unroll
          for (int k = 0; k < 8; ++k) am[k] = (v[k] > m[k]) ? code : am[k];
        }
        m = __builtin_elementwise_max(m, v);
      }
    }
    if (a.argmax) {
      uint64_t pk = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) pk |= (uint64_t)(am[k] & 0xff) << (8 * k);
      *(uint64_t*)(a.argmax + ((size_t)(f * a.CB + cb) * PH * PW + (size_t)(py * PW + px)) * 32 + oct * 8) = pk;
    }
    const u32x4 mv = __builtin_bit_cast(u32x4, m);
    float vals[8];
    unpack8(mv, vals);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s_sum += vals[k];
      s_sq = fmaf(vals[k], vals[k], s_sq);
    }
    *(u32x4*)(a.y + ((size_t)(f * a.CB + cb) * PH * PW + (size_t)(py * PW + px)) * 32 + oct * 8) = mv;
  }
  if (a.stats_out) block_stats_atomic(s_sum, s_sq, a.stats_out, f);
}

__global__ __launch_bounds__(256) void vpt_affine_kernel(VptAffineArgs a) {
  const int per_frame = a.CB * a.HW * 4;  // 16-byte items
  const int blocks_per_frame = (per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK;
  const int f = blockIdx.x / blocks_per_frame;
  const int base = (blockIdx.x - f * blocks_per_frame) * EW_PER_BLOCK + threadIdx.x;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
  const float shift = -mean * rstd;
  u32x4 xv[EW_ITEMS];
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = base + it * 256;
    if (item < per_frame) xv[it] = *(const u32x4*)(a.x + (size_t)f * per_frame * 8 + (size_t)item * 8);
  }
  float s_sum = 0.f, s_sq = 0.f;
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = base + it * 256;
    if (item >= per_frame) break;
    float v[8];
    unpack8(xv[it], v);
    int gidx;
    if (a.per_element) {
      gidx = item * 8;
    } else {
      const int cb = item / (a.HW * 4);
      gidx = cb * 32 + (item & 3) * 8;
    }
    const f32x4 g0 = *(const f32x4*)(a.gain + gidx), g1 = *(const f32x4*)(a.gain + gidx + 4);
    const f32x4 b0 = *(const f32x4*)(a.bias + gidx), b1 = *(const f32x4*)(a.bias + gidx + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = fmaf(fmaf(v[k], rstd, shift), g[k], b[k]);
      s_sum += v[k];
      s_sq = fmaf(v[k], v[k], s_sq);
    }
    *(u32x4*)(a.y + (size_t)f * per_frame * 8 + (size_t)item * 8) = pack8(v);
  }
  if (a.stats_out) block_stats_atomic(s_sum, s_sq, a.stats_out, f);
}

extern "C" int vpt_pool_launch(const VptPoolArgs* a, hipStream_t stream) {
  if ((a->H & 1) || (a->W & 1) || a->frames <= 0) return -1;
  const int per_frame = a->CB * (a->H >> 1) * (a->W >> 1) * 4;
  const long grid = (long)a->frames * ((per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK);
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_pool_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int vpt_affine_launch(const VptAffineArgs* a, hipStream_t stream) {
  if (a->frames <= 0) return -1;
  const int per_frame = a->CB * a->HW * 4;
  const long grid = (long)a->frames * ((per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK);
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_affine_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
