// Backward kernels of the behavioural-cloning step for the heads / trunk / transformer (gfx950), fp32.
//
// The BC loss is  L = -mean_{b,t}[ log pi_buttons(a_bt) + log pi_camera(a_bt) ]  (behavioural_cloning.py:107-119,
// lib/action_head.py:176-184,252-253); the reference obtains its gradient from torch autograd.  Linear layers
// reuse vpt_gemm_kernel (dgrad with W^T packed, wgrad with A = dY^T); this file holds the rest:
//
//  vpt_nll_bwd_kernel    : d/dz of -(log_softmax(z/T)[a]) summed over the two heads -> bf16 dz (GEMM operand).
//  vpt_ln_bwd_kernel     : nn.LayerNorm backward (optionally through a ReLU on its input), dgain/dbias by
//                          per-workgroup register partials -> the workgroup's row of a partial slab -> vpt_ln_bwd_finish_kernel
//                          (fixed summation order: bit-reproducible).
//  vpt_colsum_kernel     : bias gradients (column sums of a bf16 matrix).
//  vpt_attn_bwd_kernel   : backward of vpt_attn_kernel (banded attention with KV memory + rel-pos bias): per
//                          (sequence, head, 32-query tile) recompute P, then dV = P^T dO, dP = dO V^T,
//                          dS = P (dP - rowsum(P dP)), dQ = dS K / d, dK = dS^T Q / d, dR = dS B, db_nd = R^T dS.
//                          Keys are shared by up to five query tiles -> each tile's dK / dV piece goes to its own slab slot, db_nd to
//                          the workgroup's slab row; vpt_attn_bwd_finish_kernel / vpt_slab_sum add them in a fixed order (bit-reproducible);
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
      if (gb && !(a.mask_buttons && !a.mask_buttons[(size_t)row * a.nb + i])) g = (gb[i] - expf(lb[i]) * sb) * a.inv_temp * a.grad_scale;
    } else if (i < a.nb + a.nc) {
      const int j = i - a.nb;
      if (gc && !(a.mask_camera && !a.mask_camera[(size_t)row * a.nc + j])) g = (gc[j] - expf(lc[j]) * sc) * a.inv_temp * a.grad_scale;
    }
    else if (i == a.nb + a.nc && a.g_value) g = a.g_value[row] * a.grad_scale;
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

// ND4 = float4 slots per lane (D <= 256 ND4), a compile-time bound: the row lives in registers (x always, dy too when
// ND4 <= 8) for its four passes, and the per-lane column partials pg / pb are indexed statically.
template <int ND4>
__global__ __launch_bounds__(256) void vpt_ln_bwd_kernel(VptLnBwdArgs a) {
  constexpr bool KEEP_DY = ND4 <= 8;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n4 = a.D >> 2;            // float4 per row
  f32x4 pg[ND4], pb[ND4];
#pragma unroll
  for (int q = 0; q < ND4; ++q) { pg[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; pb[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float invD = 1.0f / (float)a.D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
    const int row = blockIdx.x * LNB_ROWS + w * (LNB_ROWS / 4) + rr;
    if (row >= a.M) break;
    const float* x = a.x + (size_t)row * a.D;
    const float* dy = a.dy + (size_t)row * a.D;
    f32x4 xv[ND4], dv[KEEP_DY ? ND4 : 1];
#pragma unroll
    for (int q = 0; q < ND4; ++q) {
      const int i = lane + 64 * q;
      xv[q] = (i < n4) ? *(const f32x4*)(x + 4 * i) : zero;
      if (KEEP_DY) dv[q] = (i < n4) ? *(const f32x4*)(dy + 4 * i) : zero;
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < ND4; ++q) {
      if (a.relu_in) { xv[q].x = fmaxf(xv[q].x, 0.f); xv[q].y = fmaxf(xv[q].y, 0.f); xv[q].z = fmaxf(xv[q].z, 0.f); xv[q].w = fmaxf(xv[q].w, 0.f); }
      s += (xv[q].x + xv[q].y) + (xv[q].z + xv[q].w);     // (slots past the row hold zeros)
    }
    const float mean = wave_sum(s) * invD;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < ND4; ++q) {
      if (lane + 64 * q < n4) {
        const float d0 = xv[q].x - mean, d1 = xv[q].y - mean, d2 = xv[q].z - mean, d3 = xv[q].w - mean;
        ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) * invD + VPT_NORM_EPS);
    // sums of dy*g and dy*g*xhat over the row
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < ND4; ++q) {
      const int i = lane + 64 * q;
      if (i < n4) {
        const f32x4 g = *(const f32x4*)(a.gain + 4 * i), d = KEEP_DY ? dv[q] : *(const f32x4*)(dy + 4 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (xv[q][k] - mean) * rstd, dg = d[k] * g[k];
          s1 += dg;
          s2 = fmaf(dg, xh, s2);
        }
      }
    }
#ifdef VPT_LN_DEBUG   // diagnostics build (tools/ubench/pk_hazard): every lane's partial sums and the reduced values of every row, [M][64][4]
    if (a.debug) { a.debug[((size_t)row * 64 + lane) * 4 + 0] = s1; a.debug[((size_t)row * 64 + lane) * 4 + 1] = s2; }
#endif
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
#ifdef VPT_LN_DEBUG
    if (a.debug) { a.debug[((size_t)row * 64 + lane) * 4 + 2] = s1; a.debug[((size_t)row * 64 + lane) * 4 + 3] = s2; }
#endif
#pragma unroll
    for (int q = 0; q < ND4; ++q) {
      const int i = lane + 64 * q;
      if (i < n4) {
        const f32x4 g = *(const f32x4*)(a.gain + 4 * i), d = KEEP_DY ? dv[q] : *(const f32x4*)(dy + 4 * i);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (xv[q][k] - mean) * rstd;
          float dxk = rstd * (d[k] * g[k] - s1 - xh * s2);
          if (a.relu_in && !(xv[q][k] > 0.f)) dxk = 0.f;     // max(x, 0) > 0  <=>  x > 0
          o[k] = dxk;
          pg[q][k] = fmaf(d[k], xh, pg[q][k]);
          pb[q][k] += d[k];
        }
        if (a.dx_add) {
          const f32x4 e = *(const f32x4*)(a.dx_add + (size_t)row * a.D + 4 * i);
          o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        *(f32x4*)(a.dx + (size_t)row * a.D + 4 * i) = o;
      }
    }
  }
  // column partials: every WAVE writes its sums to its own row of the partial slab (row = 4 x workgroup + wave; float4 per lane, coalesced) and
  // vpt_ln_bwd_finish_kernel adds the rows in row order -- no LDS table, no barrier, nothing that depends on the order in which waves or
  // workgroups run.  (Until round 5: LDS float atomics across the waves + one global fp32 atomic per column per workgroup.)
  float* part = a.partials + ((size_t)blockIdx.x * 4 + w) * 2 * a.D;
#pragma unroll
  for (int q = 0; q < ND4; ++q) {
    const int i = lane + 64 * q;
    if (i < n4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {      // (element stores: as 16-byte stores the partials would have to sit in aligned register quadruples -- 600 spills at ND4 = 8)
        part[4 * i + k] = pg[q][k];
        part[a.D + 4 * i + k] = pb[q][k];
      }
    }
  }
}

// dgain[c] += sum_b partials[b][0][c], dbias[c] += sum_b partials[b][1][c] over the nblocks = 4 x workgroups rows: a workgroup takes 16 columns x 16 contiguous
// row segments (b ascending inside a segment), the segments are combined by a fixed binary tree -- one summation order per column, a function of
// (nblocks, D) alone, whatever the order the workgroups above ran in.  (16 segments: 64 dependent-free loads per thread at M = 8192 instead of 256.)
__global__ __launch_bounds__(256) void vpt_ln_bwd_finish_kernel(const float* __restrict__ partials, int nblocks, int D, float* dgain, float* dbias) {
  __shared__ float seg_[16][16];
  const int lc = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + lc;                                   // col indexes the 2 D columns [dgain ; dbias]
  const int per = (nblocks + 15) >> 4, b0 = sg * per, b1 = min(b0 + per, nblocks);
  float s = 0.f;
  if (col < 2 * D) {
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += partials[(size_t)b * 2 * D + col];
  }
  seg_[sg][lc] = s;
  __syncthreads();
  if (sg == 0 && col < 2 * D) {
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = seg_[i][lc];
#pragma unroll
    for (int w = 1; w < 16; w <<= 1)
#pragma unroll
      for (int i = 0; i < 16; i += 2 * w) t[i] += t[i + w];
    if (col < D) dgain[col] += t[0];
    else dbias[col - D] += t[0];
  }
}

extern "C" int vpt_ln_bwd_launch(const VptLnBwdArgs* a, hipStream_t stream) {
  if (a->M <= 0 || (a->D & 3) || a->D > 4 * 64 * LNB_MAXD4 || !a->partials) return -1;
  const dim3 g((a->M + LNB_ROWS - 1) / LNB_ROWS), b(256);
  const int nd4 = ((a->D >> 2) + 63) >> 6;
  if (nd4 <= 4) hipLaunchKernelGGL(vpt_ln_bwd_kernel<4>, g, b, 0, stream, *a);
  else if (nd4 <= 8) hipLaunchKernelGGL(vpt_ln_bwd_kernel<8>, g, b, 0, stream, *a);
  else if (nd4 <= 12) hipLaunchKernelGGL(vpt_ln_bwd_kernel<12>, g, b, 0, stream, *a);
  else hipLaunchKernelGGL(vpt_ln_bwd_kernel<16>, g, b, 0, stream, *a);
  hipLaunchKernelGGL(vpt_ln_bwd_finish_kernel, dim3((2 * a->D + 15) / 16), b, 0, stream, (const float*)a->partials, 4 * (int)g.x, a->D, a->dgain, a->dbias);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// out[col] += sum_r x[r][col]: the rows are cut into gridDim.y slices, slice sums go to the slab partials[slice][col] and
// vpt_colsum_finish_kernel adds them in slice order (bit-reproducible; until round 5 one fp32 atomic per slice).
__global__ __launch_bounds__(256) void vpt_colsum_kernel(VptColsumArgs a) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int rows_per = (a.M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(r0 + rows_per, a.M);
  if (col >= a.N) return;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += (float)a.x[(size_t)r * a.ld + col];
  if (gridDim.y == 1) a.out[col] += s;
  else a.partials[(size_t)blockIdx.y * a.N + col] = s;
}

__global__ __launch_bounds__(256) void vpt_colsum_finish_kernel(VptColsumArgs a, int slices) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= a.N) return;
  float s = 0.f;
  for (int y = 0; y < slices; ++y) s += a.partials[(size_t)y * a.N + col];
  a.out[col] += s;
}

static int colsum_slices(int M) { return M >= 4096 ? 64 : (M >= 256 ? 16 : 1); }
extern "C" long vpt_colsum_partial_floats(int M, int N) { const int gy = colsum_slices(M); return gy > 1 ? (long)gy * N : 0; }

extern "C" int vpt_colsum_launch(const VptColsumArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->N <= 0) return -1;
  const int gy = colsum_slices(a->M);
  if (gy > 1 && !a->partials) return -1;
  hipLaunchKernelGGL(vpt_colsum_kernel, dim3((a->N + 255) / 256, gy), dim3(256), 0, stream, *a);
  if (gy > 1) hipLaunchKernelGGL(vpt_colsum_finish_kernel, dim3((a->N + 255) / 256), dim3(256), 0, stream, *a, gy);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
#define ATT_DH 128
#define ATT_QT 32
#define ATT_NK 160
#define ATT_SS 164                      // padded score row, 16-byte aligned
#define AB_P_OFF 0                                        // P  [32][ATT_SS]
#define AB_D_OFF (AB_P_OFF + ATT_QT * ATT_SS)             // dP, then dS  [32][ATT_SS]
#define AB_R_OFF (AB_D_OFF + ATT_QT * ATT_SS)
#define AB_B_OFF (AB_R_OFF + ATT_QT * 10)
#define AB_BP 133                                         // pitch of a b_nd row in LDS: odd, so that ten lanes reading one offset of ten DIFFERENT rows (the
                                                          // dR loop) hit ten banks -- at pitch maxlen = 128 they all hit one (47 % of this kernel's LDS cycles
                                                          // were bank conflicts in the round-3 PMC survey)
#define AB_PT_OFF ((AB_B_OFF + 10 * AB_BP + 3) & ~3)      // partial tiles of key tile 4: [wave][16][64]
#define AB_FLOATS (AB_PT_OFF + 4 * 16 * 64)

// Round 2: every contraction runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) with its operands taken straight from global
// memory (each element is used by exactly one lane) or from the 32 x 160 score tiles in LDS; 65 KB of LDS -> two workgroups per
// CU.  The first version staged Q, dO and the K / V slab in 151 KB of LDS and did the five contractions on the vector ALU.
//   "QK-like"  (logits, dP):  M = queries, N = keys, K = d_head; A = a query row, B = a key row, both contiguous in d.
//   "PV-like"  (dQ):          M = queries, N = d_head, K = keys;  A = a score row (LDS, contiguous in keys), B = key rows.
//   "dV-like"  (dV, dK):      M = keys,    N = d_head, K = queries; A = a score column (LDS), B = query rows.
// A lane supplies the k values 8 g + 4 hi + e, e = 0..3, of group g in every case (the k order of an MFMA step is free as long
// as both operands agree).
__global__ __launch_bounds__(256, 2) void vpt_attn_bwd_kernel(VptAttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ps = sm + AB_P_OFF;
  float* Ds = sm + AB_D_OFF;
  float* Rs = sm + AB_R_OFF;
  float* Bs = sm + AB_B_OFF;
  float* Pt = sm + AB_PT_OFF;

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y / a.heads, h = blockIdx.y - b * a.heads;
  const int q0 = blockIdx.x * ATT_QT;
  const int maxlen = a.maxlen, t = a.t, hid = a.hid;
  const size_t tok0 = (size_t)b * t;
  const float inv_dh = 1.0f / ATT_DH;
  const int jbase = q0 + 1;             // key kk of the band = row jbase + kk of [memory ; chunk]
  float* dbnd_row = a.dbnd_slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (10 * a.maxlen);

  for (int idx = tid; idx < ATT_QT * 10; idx += 256) {
    const int r = idx / 10, n = idx - r * 10;
    Rs[idx] = (q0 + r < t) ? a.qkvr[(tok0 + q0 + r) * a.ld + 3 * hid + h * 10 + n] : 0.f;
  }
  for (int idx = tid; idx < 10 * maxlen; idx += 256) { const int n = idx / maxlen; Bs[n * AB_BP + (idx - n * maxlen)] = a.b_nd[idx]; }
  __syncthreads();

  // row j of [memory ; chunk] -> its K (which = 1) / V (which = 2) row of this head, or null beyond the chunk
  auto kv_row = [&](int j, int which) -> const float* {
    if (j < maxlen) return (which == 1 ? a.kmem : a.vmem) + ((size_t)b * maxlen + j) * hid + h * ATT_DH;
    if (j - maxlen < t) return a.qkvr + (tok0 + j - maxlen) * a.ld + which * hid + h * ATT_DH;
    return nullptr;
  };
  // query-indexed rows as (wave-uniform base pointer, 32-bit element offset or -1): one scalar register per row instead of a 64-bit pointer
  // (the pointer-returning form kept 8 pointers x several unrolled iterations alive: 163 spilled SGPRs)
  const float* qbase = a.qkvr + (tok0 + q0) * a.ld + h * ATT_DH;
  const float* dobase = a.dout + (tok0 + q0) * hid + h * ATT_DH;
  const int nq_valid = t - q0;                            // queries q0 + qi with qi < nq_valid exist
  auto q_row = [&](int qi) -> const float* { return (qi < nq_valid) ? qbase + qi * a.ld : nullptr; };
  auto do_row = [&](int qi) -> const float* { return (qi < nq_valid) ? dobase + qi * hid : nullptr; };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const uint8_t* memv = a.memvalid + (size_t)b * maxlen;

  // QK-like contraction: key tiles 0..3 belong to waves 0..3 (acc), tile 4 (keys 128..159) is split over the waves by d
  // (four groups each, acc4) and summed through Pt, so every wave issues 80 MFMAs.  emit(key, r, value) receives the tile.
  auto qk_like = [&](const float* arow, int which, auto emit) __attribute__((always_inline)) {
    f32x16 acc, acc4;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc4[r] = 0.f; }
    const float* qa = arow ? arow + 4 * hi : nullptr;
    const float* kr = kv_row(jbase + w * 32 + l31, which);
    const float* kr4 = kv_row(jbase + 128 + l31, which);
    const float* kb = kr ? kr + 4 * hi : nullptr;
    const float* kb4 = kr4 ? kr4 + 4 * hi : nullptr;
#pragma unroll 8
    for (int g = 0; g < 16; ++g) {
      const f32x4 q4 = qa ? *(const f32x4*)(qa + 8 * g) : z4, k4 = kb ? *(const f32x4*)(kb + 8 * g) : z4;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, k4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, k4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, k4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, k4.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int g = 4 * w + g4;
      const f32x4 q4 = qa ? *(const f32x4*)(qa + 8 * g) : z4, k4 = kb4 ? *(const f32x4*)(kb4 + 8 * g) : z4;
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, k4.x, acc4, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, k4.y, acc4, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, k4.z, acc4, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, k4.w, acc4, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Pt[(w * 16 + r) * 64 + lane] = acc4[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) emit(w * 32 + l31, r, acc[r]);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {   // key tile 4: this wave finishes accumulator registers 4 w .. 4 w + 3
      const int r = 4 * w + r4;
      emit(128 + l31, r, (Pt[(0 * 16 + r) * 64 + lane] + Pt[(1 * 16 + r) * 64 + lane]) + (Pt[(2 * 16 + r) * 64 + lane] + Pt[(3 * 16 + r) * 64 + lane]));
    }
    __syncthreads();
  };
  // accumulator register r of a lane = row (r & 3) + 8 (r >> 2) + 4 hi of the 32 x 32 tile, column l31
#define ROW_OF(r_) (((r_) & 3) + 8 * ((r_) >> 2) + 4 * hi)

  // ---- 1. logits (as the forward kernel) -> Ps ----
  qk_like(q_row(l31), 1, [&](int kk, int r, float dot) {
    const int qi = ROW_OF(r);
    const int off = maxlen - 1 + qi - kk;  // 0 = the query itself, maxlen-1 = oldest key in the band
    const int j = jbase + kk;
    bool vis = (q0 + qi) < t && off >= 0 && off < maxlen;
    if (vis && j < maxlen) vis = memv[j] != 0;
    float sc = -3.0e38f;
    if (vis) {
      float rb = 0.f;
#pragma unroll
      for (int n = 0; n < 10; ++n) rb = fmaf(Rs[qi * 10 + n], Bs[n * AB_BP + off], rb);
      sc = dot * inv_dh + rb;
    }
    Ps[qi * ATT_SS + kk] = sc;
  });
  // ---- softmax -> normalised P in Ps (8 rows per wave) ----
  for (int r = w * 8; r < w * 8 + 8; ++r) {
    float* srow = Ps + r * ATT_SS;
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
  // ---- 2. dP = dO V^T -> Ds (its trailing barrier also publishes P) ----
  qk_like(do_row(l31), 2, [&](int kk, int r, float dot) { Ds[ROW_OF(r) * ATT_SS + kk] = dot; });
  // ---- 3. D_i = sum_k P dP;  dS = P (dP - D_i) -> Ds (0 where masked since P = 0) ----
  for (int r = w * 8; r < w * 8 + 8; ++r) {
    const float* prow = Ps + r * ATT_SS;
    float* drow = Ds + r * ATT_SS;
    const float p0 = prow[lane], p1 = prow[lane + 64], p2 = (lane + 128 < ATT_NK) ? prow[lane + 128] : 0.f;
    const float d0 = drow[lane], d1 = drow[lane + 64], d2 = (lane + 128 < ATT_NK) ? drow[lane + 128] : 0.f;
    const float Di = wave_sum(fmaf(p0, d0, fmaf(p1, d1, p2 * d2)));
    drow[lane] = p0 * (d0 - Di);
    drow[lane + 64] = p1 * (d1 - Di);
    if (lane + 128 < ATT_NK) drow[lane + 128] = p2 * (d2 - Di);
  }
  __syncthreads();

  const int dcol = w * 32 + l31;        // wave w owns d_head slice 32 w .. 32 w + 31 from here on
  const float* kmem_h = a.kmem + (size_t)b * maxlen * hid + h * ATT_DH;        // K rows of the memory / of the chunk, this head
  const float* kchunk_h = a.qkvr + tok0 * a.ld + hid + h * ATT_DH;
  // dV-like contraction: out[key][d] = scale * sum_query X[query][key] Y[query][d]; accumulated into the K / V columns of the
  // chunk's keys (memory keys are detached state: no gradient)
  auto dv_like = [&](const float* X, const float* ybase, int ystride, int which, float scale) __attribute__((always_inline)) {
    f32x16 acc[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {   // not unrolled: eight wave-uniform row pointers per iteration live in scalar registers
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qi = 8 * g + 4 * hi + e;          // this lane's query row of the k step
        y[e] = (qi < nq_valid) ? ybase[qi * ystride + dcol] : 0.f;
      }
#pragma unroll
      for (int kt = 0; kt < 5; ++kt) {
        const float* xc = X + (8 * g + 4 * hi) * ATT_SS + kt * 32 + l31;     // A: row = key, k = queries 8 g + 4 hi + e
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[0], y[0], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[ATT_SS], y[1], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[2 * ATT_SS], y[2], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[3 * ATT_SS], y[3], acc[kt], 0, 0, 0);
      }
    }
    // A key of the chunk is reached by up to five query tiles (its own and the four behind it).  Each writes its piece into a slab slot of its own --
    // slot = (this query tile) - (the key's 32-row tile), 0..4 -- and vpt_attn_bwd_finish_kernel adds the slots in slot order: bit-reproducible
    // (until round 5: fp32 atomics into dqkvr, i.e. sums in arrival order).  Rows beyond q0 + 31 lie outside the band (their pieces are exact
    // zeros) and are not written; the finish kernel knows which (row, slot) pairs exist from the same arithmetic.
    // wave-uniform base + one 32-bit per-lane index per element (80 64-bit vector addresses would not fit the register file)
    float* colbase = a.dkv_slab + tok0 * (2 * hid) + (which - 1) * hid + h * ATT_DH;
    const int slot_stride = a.B * t * 2 * hid;        // (the launcher checks 5 x this fits 31 bits)
    const int jr0 = jbase - maxlen + 4 * hi;          // chunk row of accumulator register 0 of key tile 0
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int jr = jr0 + kt * 32 + (r & 3) + 8 * (r >> 2);
        asm volatile("" : "+v"(jr));     // keep the validity test next to its store: hoisted and shared between the two calls, the 80 lane masks
                                         // were 160 scalar registers, all spilled
        if ((unsigned)jr < (unsigned)t && jr <= q0 + ATT_QT - 1)
          colbase[((int)blockIdx.x - (jr >> 5)) * slot_stride + jr * (2 * hid) + dcol] = acc[kt][r] * scale;
      }
  };
  // ---- 4. dV += P^T dO ;  5. dK += dS^T Q / d_h ----
  dv_like(Ps, dobase, hid, 2, 1.0f);
  dv_like(Ds, qbase, a.ld, 1, inv_dh);
  // ---- 6. dQ = dS K / d_h (PV-like) ----
  {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const float* pa = Ds + l31 * ATT_SS + 4 * hi;
#pragma unroll 2
    for (int g = 0; g < ATT_NK / 8; ++g) {
      const f32x4 p4 = *(const f32x4*)(pa + 8 * g);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {          // per-LANE row arithmetic (32-bit offsets from two uniform bases), no wave-uniform pointer sets
        const int j = jbase + 8 * g + 4 * hi + e;
        const float* rr = (j < maxlen) ? kmem_h + (size_t)j * hid : ((j - maxlen < t) ? kchunk_h + (size_t)(j - maxlen) * a.ld : nullptr);
        v[e] = rr ? rr[dcol] : 0.f;
      }
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.x, v[0], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.y, v[1], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.z, v[2], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.w, v[3], o, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = ROW_OF(r);
      if (q0 + qi < t) a.dqkvr[(tok0 + q0 + qi) * a.ld + h * ATT_DH + dcol] = o[r] * inv_dh;
    }
  }
#undef ROW_OF
  // ---- 7. dR[qi][n] = sum_kk dS B[n][off];  db_nd[n][off] += sum_qi dS R[qi][n] ----
  for (int idx = tid; idx < ATT_QT * 10; idx += 256) {
    const int r = idx / 10, n = idx - r * 10;
    float s = 0.f;
    for (int kk = 0; kk < ATT_NK; ++kk) {
      const int off = maxlen - 1 + r - kk;
      if (off >= 0 && off < maxlen) s = fmaf(Ds[r * ATT_SS + kk], Bs[n * AB_BP + off], s);
    }
    if (q0 + r < t) a.dqkvr[(tok0 + q0 + r) * a.ld + 3 * hid + h * 10 + n] = s;
  }
  for (int idx = tid; idx < 10 * maxlen; idx += 256) {
    const int n = idx / maxlen, off = idx - n * maxlen;
    float s = 0.f;
    for (int r = 0; r < ATT_QT; ++r) {
      const int kk = maxlen - 1 + r - off;
      if (kk >= 0 && kk < ATT_NK) s = fmaf(Ds[r * ATT_SS + kk], Rs[r * 10 + n], s);
    }
    dbnd_row[idx] = s;      // this workgroup's row of the slab; vpt_slab_sum adds the rows in row order (until round 5: an fp32 atomic per entry)
  }
}

// dqkvr[token][K and V columns] = sum over the slots that exist for the token's row, in slot order (see dv_like above).  One thread per float4.
__global__ __launch_bounds__(256) void vpt_attn_bwd_finish_kernel(VptAttnBwdArgs a) {
  const int c4n = (2 * a.hid) >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)a.B * a.t * c4n) return;
  const int tok = (int)(i / c4n), c4 = (int)(i - (long)tok * c4n);
  const int jr = tok % a.t, kt = jr >> 5;
  const size_t slot_stride = (size_t)a.B * a.t * 2 * a.hid;
  const float* p = a.dkv_slab + (size_t)tok * (2 * a.hid) + 4 * c4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) {
    const int q0 = (kt + sl) * ATT_QT;              // the query tile of slot sl
    if (q0 < a.t && jr >= q0 - a.maxlen + 1) {      // it exists and its band reaches back to this row
      const f32x4 v = *(const f32x4*)(p + sl * slot_stride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  *(f32x4*)(a.dqkvr + (size_t)tok * a.ld + a.hid + 4 * c4) = s;
}

extern "C" long vpt_attn_bwd_dkv_floats(int B, int t, int hid) { return 5L * B * t * 2 * hid; }
// rows of the db_nd slab (one per workgroup) followed by vpt_slab_sum's scratch
extern "C" long vpt_attn_bwd_dbnd_floats(int B, int t, int heads, int maxlen) {
  const long rows = (long)((t + ATT_QT - 1) / ATT_QT) * B * heads;
  return rows * 10 * maxlen + vpt_slab_sum_scratch_floats((int)rows, 10 * maxlen);
}

extern "C" int vpt_attn_bwd_launch(const VptAttnBwdArgs* a, hipStream_t stream) {
  if (a->hid != a->heads * ATT_DH || a->maxlen < 1 || a->maxlen > 129 || !a->dkv_slab || !a->dbnd_slab || (a->ld & 3)) return -1;
  if (vpt_attn_bwd_dkv_floats(a->B, a->t, a->hid) > 0x7fffffffL) return -2;       // 32-bit slab indices in the kernel
  static unsigned long long optin_done = 0;
  const size_t lds = AB_FLOATS * sizeof(float);
  if (!vpt_lds_optin((const void*)vpt_attn_bwd_kernel, (int)lds, &optin_done)) return -4;
  dim3 grid((a->t + ATT_QT - 1) / ATT_QT, a->B * a->heads);
  const long rows = (long)grid.x * grid.y;
  if (rows > 65536) return -2;
  hipLaunchKernelGGL(vpt_attn_bwd_kernel, grid, dim3(256), lds, stream, *a);
  const long n4 = (long)a->B * a->t * ((2 * a->hid) >> 2);
  hipLaunchKernelGGL(vpt_attn_bwd_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, *a);
  if (hipGetLastError() != hipSuccess) return -3;
  return vpt_slab_sum_launch(a->dbnd_slab, (int)rows, 10 * a->maxlen, 10L * a->maxlen, a->db_nd, 10 * a->maxlen, nullptr, 1,
                             a->dbnd_slab + rows * 10 * a->maxlen, stream);
}
