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
// Plain (cached) accesses here: with nontemporal ones the forward step measured 2 ms SLOWER (118.1 vs 115.9 ms, same box) -- the
// consumer of these outputs is the next convolution of the same chunk, which still finds part of them in the 256 MB MALL.
#define VPT_STREAM_PLAIN 1
#include "vpt_common.h"
#include "vpt_kernels.h"

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#define EW_ITEMS 4
#define EW_PER_BLOCK (256 * EW_ITEMS)

// All nine loads of a window are issued unconditionally from clamped coordinates (only row 2py-1 at py = 0 and column
// 2px-1 at px = 0 can fall outside: H, W even) and a padded position is zeroed after the load, so no branch sits
// between the loads and a thread keeps 9..36 of them in flight.
template <bool ARGMAX>
__global__ __launch_bounds__(256) void vpt_pool_kernel(VptPoolArgs a) {
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int per_frame = a.CB * PH * PW * 4;  // 16-byte items
  const int blocks_per_frame = (per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK;
  const int f = blockIdx.x / blocks_per_frame;
  const int base = (blockIdx.x - f * blocks_per_frame) * EW_PER_BLOCK + threadIdx.x;
  float s_sum = 0.f, s_sq = 0.f;
  u16x8 v[2][9];
  auto decode = [&](int it, int& oct, int& px, int& py, int& cb) {
    const int item = min(base + it * 256, per_frame - 1);
    oct = item & 3;
    int r = item >> 2;
    px = r % PW; r /= PW;
    py = r % PH;
    cb = r / PH;
  };
  auto load = [&](int it, u16x8* dst) {
    int oct, px, py, cb;
    decode(it, oct, px, py, cb);
    const vpt_op16* plane = a.x + ((size_t)(f * a.CB + cb) * a.H * a.W) * 32 + oct * 8;
    const int yy[3] = {max(2 * py - 1, 0), 2 * py, 2 * py + 1}, xx[3] = {max(2 * px - 1, 0), 2 * px, 2 * px + 1};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) dst[dy * 3 + dx] = VPT_LD_STREAM((const u16x8*)(plane + (size_t)(yy[dy] * a.W + xx[dx]) * 32));
  };
  load(0, v[0]);
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    if (it + 1 < EW_ITEMS) load(it + 1, v[(it + 1) & 1]);   // the next window's loads go out before this one is reduced
    __builtin_amdgcn_sched_barrier(0);
    int oct, px, py, cb;
    decode(it, oct, px, py, cb);
    const unsigned short row_keep = py ? 0xffff : 0, col_keep = px ? 0xffff : 0;
    u16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};
    u16x8 am = {15, 15, 15, 15, 15, 15, 15, 15};
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      u16x8 w = v[it & 1][q];
      if (q < 3) w &= row_keep;
      if (q % 3 == 0) w &= col_keep;
      if (ARGMAX) {  // strict > keeps the FIRST maximum in scan order (torch's rule); an all-zero window keeps 15
#pragma unroll
        for (int k = 0; k < 8; ++k) am[k] = (w[k] > m[k]) ? (unsigned short)q : am[k];
      }
      m = __builtin_elementwise_max(m, w);
    }
    if (base + it * 256 < per_frame) {
      const size_t po = ((size_t)(f * a.CB + cb) * PH * PW + (size_t)(py * PW + px)) * 32 + oct * 8;
      if (ARGMAX) {
        uint64_t pk = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) pk |= (uint64_t)(am[k] & 0xff) << (8 * k);
        VPT_ST_STREAM(pk, (uint64_t*)(a.argmax + po));
      }
      const u32x4 mv = __builtin_bit_cast(u32x4, m);
      float vals[8];
      unpack8(mv, vals);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s_sum += vals[k];
        s_sq = fmaf(vals[k], vals[k], s_sq);
      }
      VPT_ST_STREAM(mv, (u32x4*)(a.y + po));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (a.stats_out) block_stats_atomic(s_sum, s_sq, a.stats_out, f);
}

template <bool PER_ELEMENT>
__global__ __launch_bounds__(256) void vpt_affine_kernel(VptAffineArgs a) {
  const int per_frame = a.CB * a.HW * 4;  // 16-byte items
  const int blocks_per_frame = (per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK;
  const int f = blockIdx.x / blocks_per_frame;
  const int base = (blockIdx.x - f * blocks_per_frame) * EW_PER_BLOCK + threadIdx.x;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
  const float shift = -mean * rstd;
  // every load of the thread (x, gain, bias of its EW_ITEMS items; clamped, so unconditional) is issued before the first use
  u32x4 xv[EW_ITEMS];
  f32x4 g0[EW_ITEMS], g1[EW_ITEMS], b0[EW_ITEMS], b1[EW_ITEMS];
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = min(base + it * 256, per_frame - 1);
    xv[it] = VPT_LD_STREAM((const u32x4*)(a.x + (size_t)f * per_frame * 8 + (size_t)item * 8));
    const int gidx = PER_ELEMENT ? item * 8 : (item / (a.HW * 4)) * 32 + (item & 3) * 8;
    g0[it] = *(const f32x4*)(a.gain + gidx); g1[it] = *(const f32x4*)(a.gain + gidx + 4);
    b0[it] = *(const f32x4*)(a.bias + gidx); b1[it] = *(const f32x4*)(a.bias + gidx + 4);
  }
  float s_sum = 0.f, s_sq = 0.f;
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = base + it * 256;
    float v[8];
    unpack8(xv[it], v);
    const float g[8] = {g0[it].x, g0[it].y, g0[it].z, g0[it].w, g1[it].x, g1[it].y, g1[it].z, g1[it].w};
    const float b[8] = {b0[it].x, b0[it].y, b0[it].z, b0[it].w, b1[it].x, b1[it].y, b1[it].z, b1[it].w};
    if (item < per_frame) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = fmaf(fmaf(v[k], rstd, shift), g[k], b[k]);
        s_sum += v[k];
        s_sq = fmaf(v[k], v[k], s_sq);
      }
      VPT_ST_STREAM(pack8(v), (u32x4*)(a.y + (size_t)f * per_frame * 8 + (size_t)item * 8));
    }
  }
  if (a.stats_out) block_stats_atomic(s_sum, s_sq, a.stats_out, f);
}

extern "C" int vpt_pool_launch(const VptPoolArgs* a, hipStream_t stream) {
  if ((a->H & 1) || (a->W & 1) || a->frames <= 0) return -1;
  const int per_frame = a->CB * (a->H >> 1) * (a->W >> 1) * 4;
  const long grid = (long)a->frames * ((per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK);
  if (grid > 0x7fffffffL) return -2;
  if (a->argmax) hipLaunchKernelGGL(vpt_pool_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  else hipLaunchKernelGGL(vpt_pool_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int vpt_affine_launch(const VptAffineArgs* a, hipStream_t stream) {
  if (a->frames <= 0) return -1;
  const int per_frame = a->CB * a->HW * 4;
  const long grid = (long)a->frames * ((per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK);
  if (grid > 0x7fffffffL) return -2;
  if (a->per_element) hipLaunchKernelGGL(vpt_affine_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  else hipLaunchKernelGGL(vpt_affine_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------------------------------------
// GroupNorm `n` of a stack WITHOUT a pass of its own (lib/impala_cnn.py:118-121: x = self.n(x); x = block(x)).
// The producer stores Q = gain_n[c] * P (P: the pooled tensor).  With (mu_P, r_P) the frame statistics of P and kappa = r_P mu_P:
//     x[c] = n(P)[c] = r_P Q[c] + b[c],   b[c] = bias_n[c] - kappa gain_n[c]
// is never written.  Block 0's conv0 = GroupNorm(x) -> conv folds into a convolution of Q with its ordinary packed weights W':
//     out = relu( r_x r_P conv(W', Q)  +  SA[e][o] + r_x TB[e][o] - r_x kappa TG[e][o] - r_x mu_x SG[e][o] )
// (SA, SG: the layer's own edge tables; TB / TG: the same sums of W' weighted by bias_n / gain_n) and conv1 takes its residual as
// r_P Q + b[c].  (mu_x, r_x) -- the statistics of x over the frame -- follow from the PER-CHANNEL sums of Q:
//     sum x = sum_c (r_P S1[c] + HW b[c]),   sum x^2 = sum_c (r_P^2 S2[c] + 2 r_P b[c] S1[c] + HW b[c]^2).
// vpt_channel_stats_kernel reads Q once for S1 / S2 (fp32 within a workgroup, fp64 across: independent of how the batch is cut);
// vpt_nfold_coef_kernel turns them into the per-frame epilogue table kk_frame and the scalars.  What is saved: the affine pass's read AND
// write of every pooled tensor (3.25 MB per frame of the 2x model).
__global__ __launch_bounds__(256) void vpt_channel_stats_kernel(VptChannelStatsArgs a) {
  __shared__ float red[4][4][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int L = blockIdx.x;
  const int sp = L % a.split; L /= a.split;
  const int cb = L % a.CB;
  const int f = L / a.CB;
  const int per = (a.HW + a.split - 1) / a.split;
  const int p0 = sp * per, p1 = min(p0 + per, a.HW);
  const int oct = tid & 3;
  const vpt_op16* base = a.x + ((size_t)(f * a.CB + cb) * a.HW) * 32 + oct * 8;
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  for (int p = p0 + (tid >> 2); p < p1; p += 256) {       // four 16-byte loads in flight per thread
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (p + 64 * j < p1) ? VPT_LD_STREAM((const u32x4*)(base + (size_t)(p + 64 * j) * 32)) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vals[8];
      unpack8(v[j], vals);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s1[k] += vals[k]; s2[k] = fmaf(vals[k], vals[k], s2[k]); }
    }
  }
  // lanes with the same octet: lane & 3; reduce over the other 4 lane bits, then over the 4 waves
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) { s1[k] += __shfl_xor(s1[k], off, 64); s2[k] += __shfl_xor(s2[k], off, 64); }
  }
  if (lane < 4) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[w][lane][k] = s1[k]; red[w][lane][8 + k] = s2[k]; }
  }
  __syncthreads();
  if (tid < 64) {
    const int o = tid >> 4, k = tid & 15;          // octet, value (0..7 sums, 8..15 sums of squares)
    const float t = (red[0][o][k] + red[1][o][k]) + (red[2][o][k] + red[3][o][k]);
    atomicAdd(a.chs + ((size_t)f * a.CB * 32 + cb * 32 + o * 8 + (k & 7)) * 2 + (k >> 3), (double)t);
  }
}

extern "C" int vpt_channel_stats_launch(const VptChannelStatsArgs* a_in, hipStream_t stream) {
  VptChannelStatsArgs a = *a_in;
  if (a.frames <= 0 || a.CB <= 0 || a.HW <= 0) return -1;
  a.split = (a.HW + 1023) / 1024;                  // <= 1024 pixels (64 KB) per workgroup
  const long grid = (long)a.frames * a.CB * a.split;
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_channel_stats_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

__global__ __launch_bounds__(256) void vpt_nfold_coef_kernel(VptNfoldCoefArgs a) {
  __shared__ double redd[2][4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const double n_tot = (double)a.C * a.HW;
  const double mu_p = a.tot[2 * f] / n_tot;
  double var_p = a.tot[2 * f + 1] / n_tot - mu_p * mu_p;
  if (var_p < 0.0) var_p = 0.0;
  const double r_p = (double)rsqrtf((float)var_p + VPT_NORM_EPS);      // (the same fp32 rsqrt as frame_mean_rstd: the affine kernel's r_P)
  const double kappa = r_p * (double)(float)mu_p;
  double sx = 0.0, sxx = 0.0;
  for (int c = tid; c < a.C; c += 256) {
    const double q1 = a.chs[((size_t)f * a.C + c) * 2], q2 = a.chs[((size_t)f * a.C + c) * 2 + 1];
    const double b = (double)a.bias[c] - kappa * (double)a.gain[c];
    a.res_bias[(size_t)f * a.C + c] = (float)b;
    sx += r_p * q1 + (double)a.HW * b;
    sxx += r_p * r_p * q2 + 2.0 * r_p * b * q1 + (double)a.HW * b * b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off, 64); sxx += __shfl_xor(sxx, off, 64); }
  if (lane == 0) { redd[0][w] = sx; redd[1][w] = sxx; }
  __syncthreads();
  sx = (redd[0][0] + redd[0][1]) + (redd[0][2] + redd[0][3]);
  sxx = (redd[1][0] + redd[1][1]) + (redd[1][2] + redd[1][3]);
  const double mu_x = sx / n_tot;
  double var_x = sxx / n_tot - mu_x * mu_x;
  if (var_x < 0.0) var_x = 0.0;
  const float r_x = rsqrtf((float)var_x + VPT_NORM_EPS);
  if (tid == 0) {
    a.rs_frame[f] = r_x * (float)r_p;
    a.res_scale[f] = (float)r_p;
  }
  const float c_b = r_x, c_g = -r_x * (float)kappa, c_s = -r_x * (float)mu_x;
  float* kk = a.kk_frame + (size_t)f * 9 * a.CoutPad;
  for (int i = tid; i < 9 * a.CoutPad; i += 256) kk[i] = a.sa[i] + c_b * a.tb[i] + c_g * a.tg[i] + c_s * a.sg[i];
}

extern "C" int vpt_nfold_coef_launch(const VptNfoldCoefArgs* a, hipStream_t stream) {
  if (a->frames <= 0 || a->C <= 0 || a->HW <= 0 || a->CoutPad <= 0) return -1;
  hipLaunchKernelGGL(vpt_nfold_coef_kernel, dim3((unsigned)a->frames), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
