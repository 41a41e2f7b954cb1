// Backward kernels of the behavioural-cloning step for the heads / trunk / transformer (gfx950), fp32.
//
// The BC loss is  L = -mean_{b,t}[ log pi_buttons(a_bt) + log pi_camera(a_bt) ]  (behavioural_cloning.py:107-119,
// lib/action_head.py:176-184,252-253); the reference obtains its gradient from torch autograd.  Linear layers
// reuse vpt_gemm_kernel (dgrad with W^T packed, wgrad with A = dY^T); this file holds the rest:
//
//  vpt_nll_bwd_kernel    : d/dz of -(log_softmax(z/T)[a]) summed over the two heads -> bf16 dz (GEMM operand).
//  vpt_ln_bwd_kernel     : nn.LayerNorm backward (optionally through a ReLU on its input), dgain/dbias by
//                          per-workgroup register partials + one fp32 atomic per column.
//  vpt_colsum_kernel     : bias gradients (column sums of a bf16 matrix).
//  vpt_attn_bwd_kernel   : backward of vpt_attn_kernel (banded attention with KV memory + rel-pos bias): per
//                          (sequence, head, 32-query tile) recompute P, then dV = P^T dO, dP = dO V^T,
//                          dS = P (dP - rowsum(P dP)), dQ = dS K / d, dK = dS^T Q / d, dR = dS B, db_nd = R^T dS.
//                          Keys are shared by up to five query tiles -> dK / dV / db_nd accumulate with fp32 atomics;
//                          memory keys (detached state, behavioural_cloning.py:111) receive no gradient.
#include "vpt_common.h"
#include "vpt_kernels.h"

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vpt_nll_bwd_kernel(VptNllBwdArgs a) {
  const int row = blockIdx.x, tid = threadIdx.x;
  vpt_op16* dz = a.dz + (size_t)row * a.ldz;
  const float* lb = a.lp_buttons + (size_t)row * a.nb;
  const float* lc = a.lp_camera + (size_t)row * a.nc;
  const long ab = a.act_buttons[row], ac = a.act_camera[row];
  for (int i = tid; i < a.ldz; i += 256) {
    float g = 0.f;
    if (i < a.nb) g = (expf(lb[i]) - (i == ab ? 1.f : 0.f)) * a.scale;
    else if (i < a.nb + a.nc) g = (expf(lc[i - a.nb]) - ((i - a.nb) == ac ? 1.f : 0.f)) * a.scale;
    dz[i] = (vpt_op16)g;
  }
}

extern "C" int vpt_nll_bwd_launch(const VptNllBwdArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->ldz < a->nb + a->nc) return -1;
  hipLaunchKernelGGL(vpt_nll_bwd_kernel, dim3(a->M), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// Backward of lp = log_softmax(z / T) for an ARBITRARY incoming gradient g = dL/dlp (what torch autograd hands to the
// policy's outputs, lib/policy.py:271-305): dz = (g - exp(lp) * sum_j g_j) / T per head; the value column passes through.
// One workgroup per row; the row sums by wave shuffles + LDS.
__global__ __launch_bounds__(256) void vpt_heads_bwd_kernel(VptHeadsBwdArgs a) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* lb = a.lp_buttons + (size_t)row * a.nb;
  const float* lc = a.lp_camera + (size_t)row * a.nc;
  const float* gb = a.g_buttons ? a.g_buttons + (size_t)row * a.nb : nullptr;
  const float* gc = a.g_camera ? a.g_camera + (size_t)row * a.nc : nullptr;
  float sb = 0.f, sc = 0.f;
  if (gb) for (int i = tid; i < a.nb; i += 256) sb += gb[i];
  if (gc) for (int i = tid; i < a.nc; i += 256) sc += gc[i];
  sb = wave_sum(sb); sc = wave_sum(sc);
  if ((tid & 63) == 0) { red[tid >> 6] = sb; red[4 + (tid >> 6)] = sc; }
  __syncthreads();
  sb = (red[0] + red[1]) + (red[2] + red[3]);
  sc = (red[4] + red[5]) + (red[6] + red[7]);
  vpt_op16* dz = a.dz + (size_t)row * a.ldz;
  for (int i = tid; i < a.ldz; i += 256) {
    float g = 0.f;
    if (i < a.nb) {
      if (gb && !(a.mask_buttons && !a.mask_buttons[(size_t)row * a.nb + i])) g = (gb[i] - expf(lb[i]) * sb) * a.inv_temp;
    } else if (i < a.nb + a.nc) {
      const int j = i - a.nb;
      if (gc && !(a.mask_camera && !a.mask_camera[(size_t)row * a.nc + j])) g = (gc[j] - expf(lc[j]) * sc) * a.inv_temp;
    }
    else if (i == a.nb + a.nc && a.g_value) g = a.g_value[row];
    dz[i] = (vpt_op16)g;
  }
}

extern "C" int vpt_heads_bwd_launch(const VptHeadsBwdArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->ldz < a->nb + a->nc + 1) return -1;
  hipLaunchKernelGGL(vpt_heads_bwd_kernel, dim3(a->M), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// out_bf16[M][ldo] = (mask > 0 ? x : 0), columns >= N zero: ReLU gate + cast + K-padding of a GEMM A operand
__global__ __launch_bounds__(256) void vpt_gate_cast_kernel(VptGateCastArgs a) {
  const size_t total = (size_t)a.M * a.ldo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / a.ldo), col = (int)(i - (size_t)row * a.ldo);
    float v = 0.f;
    if (col < a.N) {
      v = a.x[(size_t)row * a.ldx + col];
      if (a.mask && !((float)a.mask[(size_t)row * a.ldm + col] > 0.f)) v = 0.f;
    }
    a.out[i] = (vpt_op16)v;
  }
}

extern "C" int vpt_gate_cast_launch(const VptGateCastArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->ldo < a->N) return -1;
  size_t blocks = ((size_t)a->M * a->ldo + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(vpt_gate_cast_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
#define LNB_ROWS 32  // rows per workgroup (8 per wave)
#define LNB_MAXD4 16 // D <= 4096

__global__ __launch_bounds__(256) void vpt_ln_bwd_kernel(VptLnBwdArgs a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n4 = a.D >> 2;            // float4 per row
  const int per_lane = (n4 + 63) >> 6;  // <= LNB_MAXD4
  f32x4 pg[LNB_MAXD4], pb[LNB_MAXD4];
#pragma unroll
  for (int i = 0; i < LNB_MAXD4; ++i) { pg[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; pb[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float invD = 1.0f / (float)a.D;
  for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
    const int row = blockIdx.x * LNB_ROWS + w * (LNB_ROWS / 4) + rr;
    if (row >= a.M) break;
    const float* x = a.x + (size_t)row * a.D;
    const float* dy = a.dy + (size_t)row * a.D;
    float s = 0.f;
    for (int i = lane; i < n4; i += 64) {
      f32x4 v = *(const f32x4*)(x + 4 * i);
      if (a.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) * invD;
    float ss = 0.f;
    for (int i = lane; i < n4; i += 64) {
      f32x4 v = *(const f32x4*)(x + 4 * i);
      if (a.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float rstd = rsqrtf(wave_sum(ss) * invD + VPT_NORM_EPS);
    // sums of dy*g and dy*g*xhat over the row
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < n4; i += 64) {
      f32x4 v = *(const f32x4*)(x + 4 * i);
      if (a.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const f32x4 g = *(const f32x4*)(a.gain + 4 * i), d = *(const f32x4*)(dy + 4 * i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (v[k] - mean) * rstd, dg = d[k] * g[k];
        s1 += dg;
        s2 = fmaf(dg, xh, s2);
      }
    }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
    int slot = 0;
    for (int i = lane; i < n4; i += 64, ++slot) {
      const f32x4 raw = *(const f32x4*)(x + 4 * i);
      f32x4 v = raw;
      if (a.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const f32x4 g = *(const f32x4*)(a.gain + 4 * i), d = *(const f32x4*)(dy + 4 * i);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (v[k] - mean) * rstd;
        float dxk = rstd * (d[k] * g[k] - s1 - xh * s2);
        if (a.relu_in && !(raw[k] > 0.f)) dxk = 0.f;
        o[k] = dxk;
#pragma unroll
        for (int q = 0; q < LNB_MAXD4; ++q)
          if (q == slot) { pg[q][k] = fmaf(d[k], xh, pg[q][k]); pb[q][k] += d[k]; }
      }
      if (a.dx_add) {
        const f32x4 e = *(const f32x4*)(a.dx_add + (size_t)row * a.D + 4 * i);
        o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
      }
      *(f32x4*)(a.dx + (size_t)row * a.D + 4 * i) = o;
    }
  }
  // flush the per-lane column partials: one atomic per column per wave
#pragma unroll
  for (int q = 0; q < LNB_MAXD4; ++q) {
    const int i = lane + 64 * q;
    if (q < per_lane && i < n4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(a.dgain + 4 * i + k, pg[q][k]);
        atomicAdd(a.dbias + 4 * i + k, pb[q][k]);
      }
    }
  }
}

extern "C" int vpt_ln_bwd_launch(const VptLnBwdArgs* a, hipStream_t stream) {
  if (a->M <= 0 || (a->D & 3) || a->D > 4 * 64 * LNB_MAXD4) return -1;
  hipLaunchKernelGGL(vpt_ln_bwd_kernel, dim3((a->M + LNB_ROWS - 1) / LNB_ROWS), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vpt_colsum_kernel(VptColsumArgs a) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int rows_per = (a.M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(r0 + rows_per, a.M);
  if (col >= a.N) return;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += (float)a.x[(size_t)r * a.ld + col];
  atomicAdd(a.out + col, s);
}

extern "C" int vpt_colsum_launch(const VptColsumArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->N <= 0) return -1;
  const int gy = a->M >= 4096 ? 64 : (a->M >= 256 ? 16 : 1);
  hipLaunchKernelGGL(vpt_colsum_kernel, dim3((a->N + 255) / 256, gy), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
#define ATT_DH 128
#define ATT_QT 32
#define ATT_NK 160
#define ATT_RS (ATT_DH + 4)
#define ATT_SS 161
#define AB_Q_OFF 0
#define AB_KV_OFF (ATT_QT * ATT_RS)
#define AB_S_OFF (AB_KV_OFF + ATT_NK * ATT_RS)
#define AB_DO_OFF (AB_S_OFF + ATT_QT * ATT_SS)
#define AB_R_OFF (AB_DO_OFF + ATT_QT * ATT_RS)
#define AB_B_OFF (AB_R_OFF + ATT_QT * 10)
#define AB_DB_OFF (AB_B_OFF + 10 * 129)
#define AB_RED_OFF (AB_DB_OFF + 10 * 129)
#define AB_FLOATS (AB_RED_OFF + 8 * ATT_QT)

__global__ __launch_bounds__(256) void vpt_attn_bwd_kernel(VptAttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm + AB_Q_OFF;
  float* KVs = sm + AB_KV_OFF;
  float* Ss = sm + AB_S_OFF;
  float* dOs = sm + AB_DO_OFF;
  float* Rs = sm + AB_R_OFF;
  float* Bs = sm + AB_B_OFF;
  float* dBs = sm + AB_DB_OFF;
  float* Red = sm + AB_RED_OFF;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.y / a.heads, h = blockIdx.y - b * a.heads;
  const int q0 = blockIdx.x * ATT_QT;
  const int maxlen = a.maxlen, t = a.t, hid = a.hid;
  const size_t tok0 = (size_t)b * t;
  const float inv_dh = 1.0f / ATT_DH;

  // ---- stage Q, dO, K slab, R, b_nd; zero the db_nd accumulator ----
  for (int idx = tid; idx < ATT_QT * (ATT_DH / 4); idx += 256) {
    const int r = idx >> 5, c4 = idx & 31;
    f32x4 q = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
    if (q0 + r < t) {
      q = *(const f32x4*)(a.qkvr + (tok0 + q0 + r) * a.ld + h * ATT_DH + c4 * 4);
      d = *(const f32x4*)(a.dout + (tok0 + q0 + r) * hid + h * ATT_DH + c4 * 4);
    }
    *(f32x4*)(Qs + r * ATT_RS + c4 * 4) = q;
    *(f32x4*)(dOs + r * ATT_RS + c4 * 4) = d;
  }
  for (int idx = tid; idx < ATT_NK * (ATT_DH / 4); idx += 256) {
    const int kk = idx >> 5, c4 = idx & 31;
    const int j = q0 + 1 + kk;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < maxlen) v = *(const f32x4*)(a.kmem + ((size_t)b * maxlen + j) * hid + h * ATT_DH + c4 * 4);
    else if (j - maxlen < t) v = *(const f32x4*)(a.qkvr + (tok0 + j - maxlen) * a.ld + hid + h * ATT_DH + c4 * 4);
    *(f32x4*)(KVs + kk * ATT_RS + c4 * 4) = v;
  }
  for (int idx = tid; idx < ATT_QT * 10; idx += 256) {
    const int r = idx / 10, n = idx - r * 10;
    Rs[idx] = (q0 + r < t) ? a.qkvr[(tok0 + q0 + r) * a.ld + 3 * hid + h * 10 + n] : 0.f;
  }
  for (int idx = tid; idx < 10 * maxlen; idx += 256) { Bs[idx] = a.b_nd[idx]; dBs[idx] = 0.f; }
  __syncthreads();

  const int qi = tid & 31, kg = tid >> 5;
  const bool qvalid = (q0 + qi) < t;
  // ---- logits (as the forward kernel) ----
  {
    const float* qrow = Qs + qi * ATT_RS;
    float rq[10];
#pragma unroll
    for (int n = 0; n < 10; ++n) rq[n] = Rs[qi * 10 + n];
    for (int nb = 0; nb < 5; ++nb) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      const float* k0 = KVs + (kg + 8 * (4 * nb + 0)) * ATT_RS;
      const float* k1 = KVs + (kg + 8 * (4 * nb + 1)) * ATT_RS;
      const float* k2 = KVs + (kg + 8 * (4 * nb + 2)) * ATT_RS;
      const float* k3 = KVs + (kg + 8 * (4 * nb + 3)) * ATT_RS;
#pragma unroll 8
      for (int d = 0; d < ATT_DH; d += 4) {
        const f32x4 q = *(const f32x4*)(qrow + d);
        const f32x4 x0 = *(const f32x4*)(k0 + d), x1 = *(const f32x4*)(k1 + d);
        const f32x4 x2 = *(const f32x4*)(k2 + d), x3 = *(const f32x4*)(k3 + d);
        acc[0] = fmaf(q.x, x0.x, fmaf(q.y, x0.y, fmaf(q.z, x0.z, fmaf(q.w, x0.w, acc[0]))));
        acc[1] = fmaf(q.x, x1.x, fmaf(q.y, x1.y, fmaf(q.z, x1.z, fmaf(q.w, x1.w, acc[1]))));
        acc[2] = fmaf(q.x, x2.x, fmaf(q.y, x2.y, fmaf(q.z, x2.z, fmaf(q.w, x2.w, acc[2]))));
        acc[3] = fmaf(q.x, x3.x, fmaf(q.y, x3.y, fmaf(q.z, x3.z, fmaf(q.w, x3.w, acc[3]))));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = kg + 8 * (4 * nb + u);
        const int off = maxlen - 1 + qi - kk;
        const int j = q0 + 1 + kk;
        bool vis = qvalid && off >= 0 && off < maxlen;
        if (vis && j < maxlen) vis = a.memvalid[(size_t)b * maxlen + j] != 0;
        float s = -3.0e38f;
        if (vis) {
          float rb = 0.f;
#pragma unroll
          for (int n = 0; n < 10; ++n) rb = fmaf(rq[n], Bs[n * maxlen + off], rb);
          s = acc[u] * inv_dh + rb;
        }
        Ss[qi * ATT_SS + kk] = s;
      }
    }
  }
  __syncthreads();
  // ---- softmax -> normalised P in Ss ----
  for (int r = w * 8; r < w * 8 + 8; ++r) {
    float* srow = Ss + r * ATT_SS;
    const float s0 = srow[lane], s1 = srow[lane + 64], s2 = (lane + 128 < ATT_NK) ? srow[lane + 128] : -3.0e38f;
    const float m = wave_max(fmaxf(s0, fmaxf(s1, s2)));
    const float e0 = (s0 > -1.0e38f) ? expf(s0 - m) : 0.f;
    const float e1 = (s1 > -1.0e38f) ? expf(s1 - m) : 0.f;
    const float e2 = (s2 > -1.0e38f) ? expf(s2 - m) : 0.f;
    const float tot = wave_sum(e0 + e1 + e2);
    const float inv = (tot > 0.f) ? 1.0f / tot : 0.f;
    srow[lane] = e0 * inv;
    srow[lane + 64] = e1 * inv;
    if (lane + 128 < ATT_NK) srow[lane + 128] = e2 * inv;
  }
  // ---- stage V ----
  for (int idx = tid; idx < ATT_NK * (ATT_DH / 4); idx += 256) {
    const int kk = idx >> 5, c4 = idx & 31;
    const int j = q0 + 1 + kk;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < maxlen) v = *(const f32x4*)(a.vmem + ((size_t)b * maxlen + j) * hid + h * ATT_DH + c4 * 4);
    else if (j - maxlen < t) v = *(const f32x4*)(a.qkvr + (tok0 + j - maxlen) * a.ld + 2 * hid + h * ATT_DH + c4 * 4);
    *(f32x4*)(KVs + kk * ATT_RS + c4 * 4) = v;
  }
  __syncthreads();

  // ---- dV[kk][d] += sum_qi P[qi][kk] dO[qi][d]   (thread = key group kg, 4-wide d slice dl) ----
  {
    const int dl = tid & 31;
    float acc[20][4];
#pragma unroll
    for (int n = 0; n < 20; ++n) { acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f; }
    for (int q = 0; q < ATT_QT; ++q) {
      const f32x4 d = *(const f32x4*)(dOs + q * ATT_RS + dl * 4);
      const float* prow = Ss + q * ATT_SS + kg;
#pragma unroll
      for (int n = 0; n < 20; ++n) {
        const float p = prow[8 * n];
        acc[n][0] = fmaf(p, d.x, acc[n][0]); acc[n][1] = fmaf(p, d.y, acc[n][1]);
        acc[n][2] = fmaf(p, d.z, acc[n][2]); acc[n][3] = fmaf(p, d.w, acc[n][3]);
      }
    }
#pragma unroll
    for (int n = 0; n < 20; ++n) {
      const int j = q0 + 1 + kg + 8 * n;
      if (j >= maxlen && j - maxlen < t) {
        float* dst = a.dqkvr + (tok0 + j - maxlen) * a.ld + 2 * hid + h * ATT_DH + dl * 4;
        atomicAdd(dst + 0, acc[n][0]); atomicAdd(dst + 1, acc[n][1]);
        atomicAdd(dst + 2, acc[n][2]); atomicAdd(dst + 3, acc[n][3]);
      }
    }
  }
  // ---- dP = dO V^T, D_i = sum_k P dP, dS = P (dP - D_i) ----
  float dp[20];
  {
    const float* drow = dOs + qi * ATT_RS;
    for (int nb = 0; nb < 5; ++nb) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      const float* v0 = KVs + (kg + 8 * (4 * nb + 0)) * ATT_RS;
      const float* v1 = KVs + (kg + 8 * (4 * nb + 1)) * ATT_RS;
      const float* v2 = KVs + (kg + 8 * (4 * nb + 2)) * ATT_RS;
      const float* v3 = KVs + (kg + 8 * (4 * nb + 3)) * ATT_RS;
#pragma unroll 8
      for (int d = 0; d < ATT_DH; d += 4) {
        const f32x4 q = *(const f32x4*)(drow + d);
        const f32x4 x0 = *(const f32x4*)(v0 + d), x1 = *(const f32x4*)(v1 + d);
        const f32x4 x2 = *(const f32x4*)(v2 + d), x3 = *(const f32x4*)(v3 + d);
        acc[0] = fmaf(q.x, x0.x, fmaf(q.y, x0.y, fmaf(q.z, x0.z, fmaf(q.w, x0.w, acc[0]))));
        acc[1] = fmaf(q.x, x1.x, fmaf(q.y, x1.y, fmaf(q.z, x1.z, fmaf(q.w, x1.w, acc[1]))));
        acc[2] = fmaf(q.x, x2.x, fmaf(q.y, x2.y, fmaf(q.z, x2.z, fmaf(q.w, x2.w, acc[2]))));
        acc[3] = fmaf(q.x, x3.x, fmaf(q.y, x3.y, fmaf(q.z, x3.z, fmaf(q.w, x3.w, acc[3]))));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) dp[4 * nb + u] = acc[u];
    }
    float part = 0.f;
#pragma unroll
    for (int n = 0; n < 20; ++n) part = fmaf(Ss[qi * ATT_SS + kg + 8 * n], dp[n], part);
    Red[kg * ATT_QT + qi] = part;
  }
  __syncthreads();  // all dV reads of P and all D partials are done
  {
    float Di = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) Di += Red[g * ATT_QT + qi];
#pragma unroll
    for (int n = 0; n < 20; ++n) {
      const int kk = kg + 8 * n;
      const float p = Ss[qi * ATT_SS + kk];
      Ss[qi * ATT_SS + kk] = p * (dp[n] - Di);  // dS (0 where masked since P = 0)
    }
  }
  // ---- reload K ----
  for (int idx = tid; idx < ATT_NK * (ATT_DH / 4); idx += 256) {
    const int kk = idx >> 5, c4 = idx & 31;
    const int j = q0 + 1 + kk;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < maxlen) v = *(const f32x4*)(a.kmem + ((size_t)b * maxlen + j) * hid + h * ATT_DH + c4 * 4);
    else if (j - maxlen < t) v = *(const f32x4*)(a.qkvr + (tok0 + j - maxlen) * a.ld + hid + h * ATT_DH + c4 * 4);
    *(f32x4*)(KVs + kk * ATT_RS + c4 * 4) = v;
  }
  __syncthreads();

  // ---- dQ[qi][d] = 1/d_h * sum_kk dS[qi][kk] K[kk][d]   (thread = query qi, 16-wide d slice dg) ----
  {
    const int dg = tid >> 5;
    float o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    const float* prow = Ss + qi * ATT_SS;
    const float* kcol = KVs + dg * 16;
    for (int kk = 0; kk < ATT_NK; ++kk) {
      const float p = prow[kk];
      const f32x4 v0 = *(const f32x4*)(kcol + kk * ATT_RS), v1 = *(const f32x4*)(kcol + kk * ATT_RS + 4);
      const f32x4 v2 = *(const f32x4*)(kcol + kk * ATT_RS + 8), v3 = *(const f32x4*)(kcol + kk * ATT_RS + 12);
      o[0] = fmaf(p, v0.x, o[0]); o[1] = fmaf(p, v0.y, o[1]); o[2] = fmaf(p, v0.z, o[2]); o[3] = fmaf(p, v0.w, o[3]);
      o[4] = fmaf(p, v1.x, o[4]); o[5] = fmaf(p, v1.y, o[5]); o[6] = fmaf(p, v1.z, o[6]); o[7] = fmaf(p, v1.w, o[7]);
      o[8] = fmaf(p, v2.x, o[8]); o[9] = fmaf(p, v2.y, o[9]); o[10] = fmaf(p, v2.z, o[10]); o[11] = fmaf(p, v2.w, o[11]);
      o[12] = fmaf(p, v3.x, o[12]); o[13] = fmaf(p, v3.y, o[13]); o[14] = fmaf(p, v3.z, o[14]); o[15] = fmaf(p, v3.w, o[15]);
    }
    if (qvalid) {
      float* dst = a.dqkvr + (tok0 + q0 + qi) * a.ld + h * ATT_DH + dg * 16;
#pragma unroll
      for (int i = 0; i < 16; i += 4) *(f32x4*)(dst + i) = (f32x4){o[i] * inv_dh, o[i + 1] * inv_dh, o[i + 2] * inv_dh, o[i + 3] * inv_dh};
    }
  }
  // ---- dK[kk][d] += 1/d_h * sum_qi dS[qi][kk] Q[qi][d] ----
  {
    const int dl = tid & 31;
    float acc[20][4];
#pragma unroll
    for (int n = 0; n < 20; ++n) { acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f; }
    for (int q = 0; q < ATT_QT; ++q) {
      const f32x4 d = *(const f32x4*)(Qs + q * ATT_RS + dl * 4);
      const float* prow = Ss + q * ATT_SS + kg;
#pragma unroll
      for (int n = 0; n < 20; ++n) {
        const float p = prow[8 * n];
        acc[n][0] = fmaf(p, d.x, acc[n][0]); acc[n][1] = fmaf(p, d.y, acc[n][1]);
        acc[n][2] = fmaf(p, d.z, acc[n][2]); acc[n][3] = fmaf(p, d.w, acc[n][3]);
      }
    }
#pragma unroll
    for (int n = 0; n < 20; ++n) {
      const int j = q0 + 1 + kg + 8 * n;
      if (j >= maxlen && j - maxlen < t) {
        float* dst = a.dqkvr + (tok0 + j - maxlen) * a.ld + hid + h * ATT_DH + dl * 4;
        atomicAdd(dst + 0, acc[n][0] * inv_dh); atomicAdd(dst + 1, acc[n][1] * inv_dh);
        atomicAdd(dst + 2, acc[n][2] * inv_dh); atomicAdd(dst + 3, acc[n][3] * inv_dh);
      }
    }
  }
  // ---- dR[qi][n] = sum_kk dS B[n][off];  db_nd[n][off] += sum_qi dS R[qi][n] ----
  for (int idx = tid; idx < ATT_QT * 10; idx += 256) {
    const int r = idx / 10, n = idx - r * 10;
    float s = 0.f;
    for (int kk = 0; kk < ATT_NK; ++kk) {
      const int off = maxlen - 1 + r - kk;
      if (off >= 0 && off < maxlen) s = fmaf(Ss[r * ATT_SS + kk], Bs[n * maxlen + off], s);
    }
    if (q0 + r < t) a.dqkvr[(tok0 + q0 + r) * a.ld + 3 * hid + h * 10 + n] = s;
  }
  for (int idx = tid; idx < 10 * maxlen; idx += 256) {
    const int n = idx / maxlen, off = idx - n * maxlen;
    float s = 0.f;
    for (int r = 0; r < ATT_QT; ++r) {
      const int kk = maxlen - 1 + r - off;
      if (kk >= 0 && kk < ATT_NK) s = fmaf(Ss[r * ATT_SS + kk], Rs[r * 10 + n], s);
    }
    if (s != 0.f) atomicAdd(a.db_nd + idx, s);
  }
  (void)dBs; (void)lane;
}

extern "C" int vpt_attn_bwd_launch(const VptAttnBwdArgs* a, hipStream_t stream) {
  if (a->hid != a->heads * ATT_DH || a->maxlen < 1 || a->maxlen > 129) return -1;
  static unsigned long long optin_done = 0;
  const size_t lds = AB_FLOATS * sizeof(float);
  if (!vpt_lds_optin((const void*)vpt_attn_bwd_kernel, (int)lds, &optin_done)) return -4;
  dim3 grid((a->t + ATT_QT - 1) / ATT_QT, a->B * a->heads);
  hipLaunchKernelGGL(vpt_attn_bwd_kernel, grid, dim3(256), lds, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
