// 3x3 / pad-1 convolution of the IMPALA residual CNN as an implicit GEMM on bf16 MFMA (gfx950).
//
// Replaces, per call:  FanInInitReLULayer.forward (lib/util.py:75-82) with GroupNorm(1,C) -> Conv2d(3x3,
// pad 1, no bias) -> ReLU, as used by CnnBasicBlock.conv0/conv1 (lib/impala_cnn.py:30-52) and by the
// firstconv of stacks 1..2 (lib/impala_cnn.py:86-97), plus the residual add of CnnBasicBlock.forward.
//
// Layout in HBM: activations are channel-blocked NHWC, [frame][C/32][H][W][32] bf16, so that the 18-pixel
// halo row of one 32-channel block is one contiguous 1152-byte run.  Weights are pre-packed (host side)
// as [ntile][C_in/32][tap][128 couts][32 cin] bf16 with the GroupNorm gain folded in.
//
// GroupNorm fold: conv(W, (x-mu)*rstd*g + b) with zero padding applied AFTER the norm equals
//     rstd * conv(W*g, x)  -  rstd*mu * SG[e][o]  +  SA[e][o]
// where SG/SA sum W*g / W*b over the taps that are inside the image for the pixel's edge class e
// (3 row classes x 3 column classes).  The main loop therefore streams raw bf16 activations; the
// per-frame statistics (sum, sum of squares; produced by the previous kernel's epilogue) enter only
// in the epilogue.
//
// Tiling: one workgroup (4 waves) = 16x16 output pixels x 128 output channels of one frame.
// K loop: for each 32-channel block the 18x18x32 halo tile is staged once in LDS and reused by all nine
// taps; the weight tile of three taps (one kernel row) is staged per step.  Wave tile 128 px x 64 couts
// = 4x2 MFMA 32x32x16 accumulators.  Global->register prefetch of the next step overlaps the MFMAs
// (issue early, ds_write late); two workgroups per CU hide each other's barriers.
#include "vpt_common.h"
#include "vpt_kernels.h"
#include <stdlib.h>

#define A_RS 80
#define A_BYTES (324 * A_RS)          // 25920
#define B_RS 80
#define B_BYTES (3 * 128 * B_RS)      // 30720
#define STG_F 68                      // floats per staging row (64 + 4 pad)
#define STG_WAVE (32 * STG_F * 4)     // 8704 bytes
#define KK_OFF (4 * STG_WAVE)         // 34816

__global__ __launch_bounds__(256, 2) void vpt_conv3x3_kernel(VptConv3x3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + B_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int tilesX = a.W >> 4, tilesY = a.H >> 4;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = L % a.NT; L /= a.NT;
  const int tx = L % tilesX; L /= tilesX;
  const int ty = L % tilesY;
  const int f = L / tilesY;
  const int tx0 = tx * 16, ty0 = ty * 16;
  const int NCB = a.Cin >> 5;
  const int HW = a.H * a.W;

  // ---- staging maps (fixed for the whole K loop) ----
  int a_goff[6], a_loff[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int q = tid + 256 * m;
    if (q < 1296) {
      const int hy = q / 72, rem = q - hy * 72, hx = rem >> 2, part = rem & 3;
      const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
      a_loff[m] = (hy * 18 + hx) * A_RS + part * 16;
      a_goff[m] = (y >= 0 && y < a.H && x >= 0 && x < a.W) ? (y * a.W + x) * 32 + part * 8 : -1;
    } else {
      a_loff[m] = -1;
      a_goff[m] = -1;
    }
  }
  const bf16_t* xplane = a.x + (size_t)f * NCB * HW * 32;
  const bf16_t* wbase = a.wpk + (size_t)nt * NCB * 9 * 4096 + tid * 8;
  unsigned char* bst = smem + A_BYTES + (tid >> 2) * B_RS + (tid & 3) * 16;

  u32x4 areg[6], breg[6];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // prologue: stage channel block 0 and step 0
#pragma unroll
  for (int m = 0; m < 6; ++m) areg[m] = (a_goff[m] >= 0) ? *(const u32x4*)(xplane + a_goff[m]) : zero4;
#pragma unroll
  for (int m = 0; m < 6; ++m) breg[m] = *(const u32x4*)(wbase + m * 2048);
#pragma unroll
  for (int m = 0; m < 6; ++m)
    if (a_loff[m] >= 0) *(u32x4*)(smem + a_loff[m]) = areg[m];
#pragma unroll
  for (int m = 0; m < 6; ++m) *(u32x4*)(bst + m * 64 * B_RS) = breg[m];
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const unsigned char* aL = smem + ((wm * 8 + (l31 >> 4)) * 18 + (l31 & 15)) * A_RS + hi * 16;
  const unsigned char* bL = smem + A_BYTES + (wn * 64 + l31) * B_RS + hi * 16;

  const int ncb_run = (a.ablate == 2) ? 0 : NCB;
  for (int cb = 0; cb < ncb_run; ++cb) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int s = cb * 3 + dy;
      const bool more = (s + 1 < NCB * 3);
      const bool nextA = (dy == 2) && (cb + 1 < NCB);
      if (more) {
        const bf16_t* wp = wbase + (size_t)(s + 1) * 12288;
#pragma unroll
        for (int m = 0; m < 6; ++m) breg[m] = *(const u32x4*)(wp + m * 2048);
      }
      if (nextA) {
        const bf16_t* xp = xplane + (size_t)(cb + 1) * HW * 32;
#pragma unroll
        for (int m = 0; m < 6; ++m) areg[m] = (a_goff[m] >= 0) ? *(const u32x4*)(xp + a_goff[m]) : zero4;
      }
      // ---- 3 taps x 2 k16-steps x (4x2) MFMA ----
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 af[4], bf[2];
#pragma unroll
          for (int m = 0; m < 4; ++m)
            af[m] = *(const bf16x8*)(aL + (dy * 18 + dx) * A_RS + m * (2 * 18 * A_RS) + ks * 32);
#pragma unroll
          for (int n = 0; n < 2; ++n) bf[n] = *(const bf16x8*)(bL + dx * (128 * B_RS) + n * (32 * B_RS) + ks * 32);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
      }
      __syncthreads();
      if (more) {
#pragma unroll
        for (int m = 0; m < 6; ++m) *(u32x4*)(bst + m * 64 * B_RS) = breg[m];
      }
      if (nextA) {
#pragma unroll
        for (int m = 0; m < 6; ++m)
          if (a_loff[m] >= 0) *(u32x4*)(smem + a_loff[m]) = areg[m];
      }
      __syncthreads();
    }
  }

  // ---------------- epilogue ----------------
  if (a.ablate == 1) {  // profiling: keep the accumulators live, skip the epilogue
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[m][n][r];
    if (t == 12345.678f) a.y[0] = (vpt_bf16)t;
    return;
  }
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
  float* kk = (float*)(smem + KK_OFF);
  for (int idx = tid; idx < 9 * 128; idx += 256) {
    const int e = idx >> 7, o = nt * 128 + (idx & 127);
    kk[idx] = a.edge_sa[e * a.CoutPad + o] - rstd * mean * a.edge_sg[e * a.CoutPad + o];
  }
  __syncthreads();

  float* stg = (float*)(smem + w * STG_WAVE);
  const float* kkw = kk + wn * 64 + l31;
  const int n0 = nt * 128 + wn * 64;
  const int CB_out = a.Cout >> 5;
  float s_sum = 0.f, s_sq = 0.f;

#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int y = ty0 + wm * 8 + 2 * m + (i >> 4);
        const int x = tx0 + (i & 15);
        const int ey = (y == 0) ? 0 : ((y == a.H - 1) ? 2 : 1);
        const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
        float v = fmaf(rstd, acc[m][n2][r], kkw[(ey * 3 + ex) * 128 + n2 * 32]);
        v = fmaxf(v, 0.f);
        stg[i * STG_F + n2 * 32 + l31] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int item = lane + 64 * it;
      const int p = item >> 3, oc = item & 7;
      const int cg = n0 + oc * 8;
      const f32x4 v0 = *(const f32x4*)(stg + p * STG_F + oc * 8);
      const f32x4 v1 = *(const f32x4*)(stg + p * STG_F + oc * 8 + 4);
      if (cg < a.Cout) {
        const int y = ty0 + wm * 8 + 2 * m + (p >> 4);
        const int x = tx0 + (p & 15);
        const size_t off = ((size_t)(f * CB_out + (cg >> 5)) * HW + (size_t)(y * a.W + x)) * 32 + (cg & 31);
        float vals[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (a.res) {
          const u32x4 rr = *(const u32x4*)(a.res + off);
          float rf[8];
          unpack8(rr, rf);
#pragma unroll
          for (int k = 0; k < 8; ++k) vals[k] += rf[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s_sum += vals[k];
          s_sq = fmaf(vals[k], vals[k], s_sq);
        }
        *(u32x4*)(a.y + off) = pack8(vals);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (a.stats_out) {
    s_sum = wave_sum(s_sum);
    s_sq = wave_sum(s_sq);
    if (lane == 0) {
      atomicAdd(a.stats_out + 2 * f, (double)s_sum);
      atomicAdd(a.stats_out + 2 * f + 1, (double)s_sq);
    }
  }
}

extern "C" int vpt_conv3x3_launch(const VptConv3x3Args* a_in, hipStream_t stream) {
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("VPT_CONV_ABLATE"); ablate = e ? atoi(e) : 0; }
  VptConv3x3Args a_copy = *a_in;
  a_copy.ablate = ablate;
  const VptConv3x3Args* a = &a_copy;
  if ((a->H & 15) || (a->W & 15) || (a->Cin & 31) || (a->Cout & 31) || a->frames <= 0) return -1;
  const long grid = (long)a->frames * (a->H >> 4) * (a->W >> 4) * a->NT;
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_conv3x3_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
