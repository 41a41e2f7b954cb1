// Backward of the HBM-bound pieces of the IMPALA CNN (gfx950): frame-wide affine norms, max-pool, and the
// per-element preparation of a normed conv layer's backward.  All tensors bf16 channel-blocked
// [frame][C/32][H][W][32]; reductions in fp32 registers -> LDS -> one fp32/fp64 atomic per workgroup.
//
//  vpt_affine_bwd_reduce / _apply : backward of y = (x - mu_f) rstd_f g + b with whole-frame statistics
//        (CnnDownStack.n = GroupNorm(1,C), lib/impala_cnn.py:99-100,118-119; ImpalaCNN.dense's LayerNorm,
//        lib/impala_cnn.py:177-184):  dx = rstd (dy g - mean_f(dy g) - xhat mean_f(dy g xhat)).
//  vpt_affine_bwd_elem            : dgain / dbias of the per-element (65536-wide) variant, reduced over frames.
//  vpt_pool_bwd                   : F.max_pool2d(3, 2, 1) backward (lib/impala_cnn.py:117) with torch's tie rule
//        (first maximum in window scan order keeps the gradient).
//  vpt_conv_bwd_prep              : for GN -> conv3x3 -> ReLU (+res) with the GroupNorm folded into the epilogue
//        (vpt_conv3x3.hip):  dz = dY * [v > 0],  dacc = rstd_f dz  (operand of the dgrad / wgrad convolutions),
//        T1_f = sum dz (v - SA[e,o]),  T2_f = sum dz SG[e,o]  (gradients w.r.t. rstd_f, mu_f),
//        dSA[e,o] = sum dz,  dSG[e,o] = sum dz (-rstd_f mu_f)  (gradients of the edge tables -> dW, dgain, dbias).
#include "vpt_common.h"
#include "vpt_kernels.h"

#define EW_ITEMS 4
#define EW_PER_BLOCK (256 * EW_ITEMS)

__device__ __forceinline__ void block_sum2_atomic_f64(float a, float b, double* dst) {
  __shared__ float red_[8];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red_[w] = a; red_[4 + w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(dst, (double)((red_[0] + red_[1]) + (red_[2] + red_[3])));
    atomicAdd(dst + 1, (double)((red_[4] + red_[5]) + (red_[6] + red_[7])));
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// pass 1: AB[f] += (sum dy g, sum dy g xhat); per-channel dgain[c] += sum dy xhat, dbias[c] += sum dy
__global__ __launch_bounds__(256) void vpt_affine_bwd_reduce_kernel(VptAffineBwdArgs a) {
  __shared__ float chan_[2 * 512];
  const int per_frame = a.CB * a.HW * 4;
  const int blocks_per_frame = (per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK;
  const int f = blockIdx.x / blocks_per_frame;
  const int base = (blockIdx.x - f * blocks_per_frame) * EW_PER_BLOCK + threadIdx.x;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
  const int C = a.CB * 32;
  if (!a.per_element)
    for (int i = threadIdx.x; i < 2 * C; i += 256) chan_[i] = 0.f;
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = base + it * 256;
    if (item >= per_frame) break;
    const size_t off = (size_t)f * per_frame * 8 + (size_t)item * 8;
    float x[8], dy[8];
    unpack8(*(const u32x4*)(a.x + off), x);
    unpack8(*(const u32x4*)(a.dy + off), dy);
    const int gidx = a.per_element ? item * 8 : (item / (a.HW * 4)) * 32 + (item & 3) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (x[k] - mean) * rstd, dg = dy[k] * a.gain[gidx + k];
      s1 += dg;
      s2 = fmaf(dg, xh, s2);
      if (!a.per_element) {
        atomicAdd(&chan_[gidx + k], dy[k] * xh);
        atomicAdd(&chan_[C + gidx + k], dy[k]);
      }
    }
  }
  block_sum2_atomic_f64(s1, s2, a.ab + 2 * f);
  if (!a.per_element) {
    for (int i = threadIdx.x; i < C; i += 256) {
      if (chan_[i] != 0.f) atomicAdd(a.dgain + i, chan_[i]);
      if (chan_[C + i] != 0.f) atomicAdd(a.dbias + i, chan_[C + i]);
    }
  }
}

// pass 2: dx = rstd (dy g - A/n - xhat B/n) [+ dx_add]
__global__ __launch_bounds__(256) void vpt_affine_bwd_apply_kernel(VptAffineBwdArgs a) {
  const int per_frame = a.CB * a.HW * 4;
  const int blocks_per_frame = (per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK;
  const int f = blockIdx.x / blocks_per_frame;
  const int base = (blockIdx.x - f * blocks_per_frame) * EW_PER_BLOCK + threadIdx.x;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
  const float A = (float)(a.ab[2 * f] * a.inv_count), B = (float)(a.ab[2 * f + 1] * a.inv_count);
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = base + it * 256;
    if (item >= per_frame) break;
    const size_t off = (size_t)f * per_frame * 8 + (size_t)item * 8;
    float x[8], dy[8], o[8];
    unpack8(*(const u32x4*)(a.x + off), x);
    unpack8(*(const u32x4*)(a.dy + off), dy);
    const int gidx = a.per_element ? item * 8 : (item / (a.HW * 4)) * 32 + (item & 3) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (x[k] - mean) * rstd;
      o[k] = rstd * (dy[k] * a.gain[gidx + k] - A - xh * B);
    }
    if (a.dx_add) {
      float e[8];
      unpack8(*(const u32x4*)(a.dx_add + off), e);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += e[k];
    }
    *(u32x4*)(a.dx + off) = pack8(o);
  }
}

// per-element gain: dgain[i] += sum_f dy xhat, dbias[i] += sum_f dy  (thread = 8 elements, grid.y = frame chunks)
__global__ __launch_bounds__(256) void vpt_affine_bwd_elem_kernel(VptAffineBwdArgs a) {
  const int per_frame = a.CB * a.HW * 4;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= per_frame) return;
  const int fper = (a.frames + gridDim.y - 1) / gridDim.y;
  const int f0 = blockIdx.y * fper, f1 = min(f0 + fper, a.frames);
  float dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, db[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int f = f0; f < f1; ++f) {
    float mean, rstd;
    frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
    const size_t off = (size_t)f * per_frame * 8 + (size_t)item * 8;
    float x[8], dy[8];
    unpack8(*(const u32x4*)(a.x + off), x);
    unpack8(*(const u32x4*)(a.dy + off), dy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dg[k] = fmaf(dy[k], (x[k] - mean) * rstd, dg[k]);
      db[k] += dy[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    atomicAdd(a.dgain + item * 8 + k, dg[k]);
    atomicAdd(a.dbias + item * 8 + k, db[k]);
  }
}

extern "C" int vpt_affine_bwd_launch(const VptAffineBwdArgs* a, int pass, hipStream_t stream) {
  if (a->frames <= 0 || (!a->per_element && a->CB * 32 > 512)) return -1;
  const int per_frame = a->CB * a->HW * 4;
  const long grid = (long)a->frames * ((per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK);
  if (grid > 0x7fffffffL) return -2;
  if (pass == 1) hipLaunchKernelGGL(vpt_affine_bwd_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  else if (pass == 2) hipLaunchKernelGGL(vpt_affine_bwd_apply_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  else {
    const int gy = a->frames >= 512 ? 32 : (a->frames >= 16 ? 8 : 1);
    hipLaunchKernelGGL(vpt_affine_bwd_elem_kernel, dim3((per_frame + 255) / 256, gy), dim3(256), 0, stream, *a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void vpt_pool_bwd_kernel(VptPoolBwdArgs a) {
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int per_frame = a.CB * a.H * a.W * 4;
  const long item_g = (long)blockIdx.x * 256 + threadIdx.x;
  if (item_g >= (long)a.frames * per_frame) return;
  const int f = (int)(item_g / per_frame);
  int r = (int)(item_g - (long)f * per_frame);
  const int oct = r & 3; r >>= 2;
  const int x = r % a.W; r /= a.W;
  const int y = r % a.H;
  const int cb = r / a.H;
  const vpt_bf16* pre = a.pre + ((size_t)(f * a.CB + cb) * a.H * a.W) * 32 + oct * 8;
  const size_t pplane = ((size_t)(f * a.CB + cb) * PH * PW) * 32 + oct * 8;
  const u16x8 mine = *(const u16x8*)(pre + (size_t)(y * a.W + x) * 32);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // windows (py, px) that contain (y, x):  2p-1 <= coord <= 2p+1
  const int py_lo = y >> 1, py_hi = (y + 1) >> 1, px_lo = x >> 1, px_hi = (x + 1) >> 1;
  for (int py = py_lo; py <= py_hi; ++py) {
    if (py >= PH) continue;
    for (int px = px_lo; px <= px_hi; ++px) {
      if (px >= PW) continue;
      const u16x8 pm = *(const u16x8*)(a.pooled + pplane + (size_t)(py * PW + px) * 32);
      float d[8];
      unpack8(*(const u32x4*)(a.dpooled + pplane + (size_t)(py * PW + px) * 32), d);
      // is (y, x) the FIRST position of the window (row-major scan) whose value equals the maximum?
      bool first_max[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) first_max[k] = (mine[k] == pm[k]);
      for (int wy = 2 * py - 1; wy <= 2 * py + 1; ++wy) {
        if (wy < 0 || wy >= a.H) continue;
        for (int wx = 2 * px - 1; wx <= 2 * px + 1; ++wx) {
          if (wx < 0 || wx >= a.W) continue;
          if (wy > y || (wy == y && wx >= x)) continue;  // only earlier positions
          const u16x8 o = *(const u16x8*)(pre + (size_t)(wy * a.W + wx) * 32);
#pragma unroll
          for (int k = 0; k < 8; ++k) first_max[k] = first_max[k] && (o[k] != pm[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += first_max[k] ? d[k] : 0.f;
    }
  }
  *(u32x4*)(a.dpre + ((size_t)(f * a.CB + cb) * a.H * a.W + (size_t)(y * a.W + x)) * 32 + oct * 8) = pack8(acc);
}

extern "C" int vpt_pool_bwd_launch(const VptPoolBwdArgs* a, hipStream_t stream) {
  if ((a->H & 1) || (a->W & 1) || a->frames <= 0) return -1;
  const long items = (long)a->frames * a->CB * a->H * a->W * 4;
  const long grid = (items + 255) / 256;
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_pool_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// One workgroup = CBP_PIX consecutive pixels of one (frame, 32-channel block).
#define CBP_PIX 1024

__global__ __launch_bounds__(256) void vpt_conv_bwd_prep_kernel(VptConvBwdPrepArgs a) {
  __shared__ float tab_[2 * 9 * 32];  // dSA / dSG partials of this block's 32 channels
  const int HW = a.H * a.W;
  const int chunks = (HW + CBP_PIX - 1) / CBP_PIX;
  int L = blockIdx.x;
  const int chunk = L % chunks; L /= chunks;
  const int cb = L % a.CB;
  const int f = L / a.CB;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
  const float nrm = -rstd * mean;
  for (int i = threadIdx.x; i < 2 * 9 * 32; i += 256) tab_[i] = 0.f;
  __syncthreads();
  const int oct = threadIdx.x & 3;
  const int c0 = cb * 32 + oct * 8;
  const size_t plane = ((size_t)(f * a.CB + cb) * HW) * 32 + oct * 8;
  float t1 = 0.f, t2 = 0.f;
  for (int pp = threadIdx.x >> 2; pp < CBP_PIX; pp += 64) {
    const int p = chunk * CBP_PIX + pp;
    if (p >= HW) break;
    const int y = p / a.W, x = p - y * a.W;
    const int ey = (y == 0) ? 0 : ((y == a.H - 1) ? 2 : 1);
    const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
    const int e = ey * 3 + ex;
    const size_t off = plane + (size_t)p * 32;
    float dy[8], v[8], o[8];
    unpack8(*(const u32x4*)(a.dy + off), dy);
    unpack8(*(const u32x4*)(a.y + off), v);
    if (a.res) {
      float rr[8];
      unpack8(*(const u32x4*)(a.res + off), rr);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] -= rr[k];
    }
    const float* sa = a.edge_sa + e * a.CoutPad + c0;
    const float* sg = a.edge_sg + e * a.CoutPad + c0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dz = (v[k] > 0.f) ? dy[k] : 0.f;
      o[k] = dz * rstd;
      t1 = fmaf(dz, v[k] - sa[k], t1);
      t2 = fmaf(dz, sg[k], t2);
      if (dz != 0.f) {
        atomicAdd(&tab_[e * 32 + oct * 8 + k], dz);
        atomicAdd(&tab_[9 * 32 + e * 32 + oct * 8 + k], dz * nrm);
      }
    }
    *(u32x4*)(a.dacc + off) = pack8(o);
  }
  block_sum2_atomic_f64(t1, t2, a.t12 + 2 * f);
  for (int i = threadIdx.x; i < 9 * 32; i += 256) {
    const int e = i >> 5, c = cb * 32 + (i & 31);
    if (tab_[i] != 0.f) atomicAdd(a.d_sa + e * a.CoutPad + c, tab_[i]);
    if (tab_[9 * 32 + i] != 0.f) atomicAdd(a.d_sg + e * a.CoutPad + c, tab_[9 * 32 + i]);
  }
}

extern "C" int vpt_conv_bwd_prep_launch(const VptConvBwdPrepArgs* a, hipStream_t stream) {
  if (a->frames <= 0) return -1;
  const int HW = a->H * a->W;
  const long grid = (long)a->frames * a->CB * ((HW + CBP_PIX - 1) / CBP_PIX);
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_conv_bwd_prep_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
