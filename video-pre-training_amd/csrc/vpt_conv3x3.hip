// 3x3 / pad-1 convolution of the IMPALA residual CNN as an implicit GEMM on bf16 MFMA (gfx950).
//
// Replaces, per call:  FanInInitReLULayer.forward (lib/util.py:75-82) with GroupNorm(1,C) -> Conv2d(3x3,
// pad 1, no bias) -> ReLU, as used by CnnBasicBlock.conv0/conv1 (lib/impala_cnn.py:30-52) and by the
// firstconv of stacks 1..2 (lib/impala_cnn.py:86-97), plus the residual add of CnnBasicBlock.forward.
//
// Layout in HBM: activations are channel-blocked NHWC, [frame][C/32][H][W][32] bf16, so that the 18-pixel
// halo row of one 32-channel block is one contiguous 1152-byte run.  Weights are pre-packed (host side)
// as [ntile][C_in/32][tap][128 couts][32 cin] bf16 with the GroupNorm gain folded in and the four 16-byte
// chunks of every 64-byte row XOR-swizzled by ((cout >> 2) & 3), i.e. already in their LDS image.
//
// GroupNorm fold: conv(W, (x-mu)*rstd*g + b) with zero padding applied AFTER the norm equals
//     rstd * conv(W*g, x)  -  rstd*mu * SG[e][o]  +  SA[e][o]
// where SG/SA sum W*g / W*b over the taps that are inside the image for the pixel's edge class e
// (3 row classes x 3 column classes).  The main loop therefore streams raw bf16 activations; the
// per-frame statistics (sum, sum of squares; produced by the previous kernel's epilogue) enter only
// in the epilogue.
//
// Tiling: one workgroup (4 waves) = 16x16 output pixels x 128 output channels of one frame; wave tile
// 128 px x 64 couts = 4x2 MFMA 32x32x16 accumulators.  K loop: for each 32-channel block the 18x18x32
// halo tile is register-staged into LDS once (zero-filled outside the image) and reused by all nine taps;
// the weight tile of one kernel row (3 taps, 24 KB) is DMA'd global->LDS (global_load_lds, no VGPRs, no
// ds_write) into a double buffer one step ahead, so a step costs ONE barrier.  round-1 profile: the first
// version of this kernel was LDS-bound (SQ_LDS_IDX_ACTIVE > MFMA busy, 39 % of it bank conflicts); hence
//   - the weight image is swizzled so ds_read_b128 of B fragments is conflict-free on 64-byte rows,
//   - the 32 rows of an MFMA M-subtile map to pixels so that every hardware 16-lane ds_read_b128 group
//     covers 16 consecutive pixels of ONE image row (80-byte pixel stride -> 16 distinct bank slots),
//   - halo ds_write_b128 are ordered so each 8-lane group hits 8 distinct 16-byte slots.
// Two workgroups per CU (78.6 KB LDS, <= 256 VGPRs) hide each other's barriers and epilogues.
#include "vpt_common.h"
#include "vpt_kernels.h"
#include <stdlib.h>

#define A_RS 80
#define A_BYTES (324 * A_RS)            // 25920
#define B_BYTES (3 * 128 * 64)          // 24576 per buffer (unpadded, swizzled)
#define KK_OFF (A_BYTES + 2 * B_BYTES)  // 75072
#define KK_BYTES (9 * 128 * 4)          // 4608
#define SMEM_BYTES (KK_OFF + KK_BYTES + 32)  // 79712 (+32: block stats reduction)

// MFMA M-subtile row i (0..31) -> pixel (row 0/1, col 0..15) of a 2x16 patch.  Rows are swapped for columns
// 4..11 so that each 16-lane ds_read_b128 group {0-3,12-15,20-27} / {4-11,16-19,28-31} stays in one image row.
__device__ __forceinline__ int sub_row(int i) { return ((i >> 4) ^ (i >> 2) ^ (i >> 3)) & 1; }

__global__ __launch_bounds__(256, 2) void vpt_conv3x3_kernel(VptConv3x3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int nsteps = NCB * 3;

  // ---- halo staging map: chunk q -> (pixel P = 8*(q>>5) + (q&7), part = (q>>3)&3) ----
  int a_goff[6], a_loff[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int q = tid + 256 * m;
    const int P = ((q >> 5) << 3) + (q & 7), part = (q >> 3) & 3;
    if (P < 324) {
      const int hy = P / 18, hx = P - hy * 18;
      const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
      a_loff[m] = P * A_RS + part * 16;
      a_goff[m] = (y >= 0 && y < a.H && x >= 0 && x < a.W) ? (y * a.W + x) * 32 + part * 8 : -1;
    } else {
      a_loff[m] = -1;
      a_goff[m] = -1;
    }
  }
  const bf16_t* xplane = a.x + (size_t)f * NCB * HW * 32;
  // weight DMA: wave w moves pieces (4*m + w), m = 0..5, of the 24 KB step tile; lane = 16-byte chunk
  const bf16_t* wbase = a.wpk + (size_t)nt * NCB * 9 * 4096 + (size_t)(w * 64 + lane) * 8;
  unsigned char* bdst = smem + A_BYTES + w * 1024;

#define ISSUE_B(step_, buf_)                                                                              \
  do {                                                                                                    \
    const bf16_t* wp_ = wbase + (size_t)(step_) * 12288;                                                  \
    _Pragma("unroll") for (int m_ = 0; m_ < 6; ++m_)                                                      \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp_ + m_ * 2048),  \
                                       (__attribute__((address_space(3))) void*)(bdst + (buf_) * B_BYTES + m_ * 4096), \
                                       16, 0, 0);                                                         \
  } while (0)

  u32x4 areg[6];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- prologue: weights of step 0 (DMA), halo of channel block 0, epilogue constant table ----
  ISSUE_B(0, 0);
#pragma unroll
  for (int m = 0; m < 6; ++m) areg[m] = (a_goff[m] >= 0) ? *(const u32x4*)(xplane + a_goff[m]) : zero4;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
  {
    float* kk = (float*)(smem + KK_OFF);
    for (int idx = tid; idx < 9 * 128; idx += 256) {
      const int e = idx >> 7, o = nt * 128 + (idx & 127);
      kk[idx] = a.edge_sa[e * a.CoutPad + o] - rstd * mean * a.edge_sg[e * a.CoutPad + o];
    }
  }
#pragma unroll
  for (int m = 0; m < 6; ++m)
    if (a_loff[m] >= 0) *(u32x4*)(smem + a_loff[m]) = areg[m];
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // fragment base addresses
  const unsigned char* aL = smem + ((wm * 8 + sub_row(l31)) * 18 + (l31 & 15)) * A_RS + hi * 16;
  const int bsw = (l31 >> 2) & 3;
  const unsigned char* bL0 = smem + A_BYTES + (wn * 64 + l31) * 64 + (((0 + hi) ^ bsw) << 4);  // ks = 0
  const unsigned char* bL1 = smem + A_BYTES + (wn * 64 + l31) * 64 + (((2 + hi) ^ bsw) << 4);  // ks = 1

  const int ncb_run = (a.ablate == 2) ? 0 : NCB;  // profiling: ablate bit 2 (4) = no weight DMA in the loop, bit 3 (8) = no halo reloads
  for (int cb = 0; cb < ncb_run; ++cb) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int s = cb * 3 + dy;
      const int buf = s & 1;
      const bool more = (s + 1 < nsteps) && !(a.ablate & 4);
      const bool nextA = (dy == 2) && (cb + 1 < NCB);
      if (more) ISSUE_B(s + 1, buf ^ 1);
      // halo of the next channel block: issued three steps (one whole block) ahead of its ds_write so that the
      // HBM latency is covered by ~150 MFMAs per wave; the registers are free now that B is DMA'd
      if ((dy == 0) && (cb + 1 < NCB) && !(a.ablate & 8)) {
        const bf16_t* xp = xplane + (size_t)(cb + 1) * HW * 32;
#pragma unroll
        for (int m = 0; m < 6; ++m) areg[m] = (a_goff[m] >= 0) ? *(const u32x4*)(xp + a_goff[m]) : zero4;
      }
      // ---- 3 taps x 2 k16-steps x (4x2) MFMA, software-pipelined: the six fragments of group g+1 are read
      //      from LDS while the eight MFMAs of group g issue (the compiler otherwise reuses one fragment
      //      register set and exposes the ds_read latency before every pair of MFMAs) ----
      const unsigned char* bB0 = bL0 + buf * B_BYTES;
      const unsigned char* bB1 = bL1 + buf * B_BYTES;
      bf16x8 fa[2][4], fb[2][2];
#define LOAD_FRAGS(g_, slot_)                                                                             \
  do {                                                                                                    \
    const int dx_ = (g_) >> 1, ks_ = (g_) & 1;                                                            \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                                      \
      fa[slot_][m_] = *(const bf16x8*)(aL + (dy * 18 + dx_) * A_RS + m_ * (2 * 18 * A_RS) + ks_ * 32);   \
    _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)                                                      \
      fb[slot_][n_] = *(const bf16x8*)((ks_ ? bB1 : bB0) + dx_ * (128 * 64) + n_ * (32 * 64));            \
  } while (0)
      LOAD_FRAGS(0, 0);
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        const int cur = g & 1;
        if (g < 5 && !(a.ablate & 16)) LOAD_FRAGS(g + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this group's MFMAs (distinct registers)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][n], fa[cur][m], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef LOAD_FRAGS
      if (!(a.ablate & 32)) __syncthreads();  // all waves done with A / B[buf]; the DMA into B[buf^1] has landed (vmcnt(0) before the barrier)
      if (nextA) {
#pragma unroll
        for (int m = 0; m < 6; ++m)
          if (a_loff[m] >= 0) *(u32x4*)(smem + a_loff[m]) = areg[m];
        __syncthreads();
      }
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

  // Operands are SWAPPED in the MFMA (weights = A rows, pixels = B columns), so a lane holds ONE pixel
  // (column l31 of the subtile) and, per accumulator, four groups of 4 consecutive output channels
  // (rows (r&3) + 8*(r>>2) + 4*hi): the epilogue is lane-local -- 16-byte reads of the constant table,
  // 8-byte residual loads and 8-byte bf16 stores straight from the accumulator layout, no LDS round trip.
  const int CB_out = a.Cout >> 5;
  const int cb0 = nt * 4 + wn * 2;                 // 32-channel block of n2 = 0
  const bool nvalid[2] = {(cb0 + 0) < CB_out, (cb0 + 1) < CB_out};
  // per M-subtile: this lane's pixel, its element offset in block cb0, and its edge-class row of the table
  size_t poff[4];
  int eoff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int y = ty0 + wm * 8 + 2 * m + sub_row(l31);
    const int x = tx0 + (l31 & 15);
    const int ey = (y == 0) ? 0 : ((y == a.H - 1) ? 2 : 1);
    const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
    eoff[m] = (ey * 3 + ex) * 128 + wn * 64 + 4 * hi;
    poff[m] = ((size_t)(f * CB_out + cb0) * HW + (size_t)(y * a.W + x)) * 32 + 4 * hi;
  }
  const size_t nstep = (size_t)HW * 32;            // next 32-channel block
  const float* kk = (const float*)(smem + KK_OFF);
  float s_sum = 0.f, s_sq = 0.f;

#pragma unroll
  for (int m = 0; m < 4; ++m) {
    // residual for this subtile: 8 x 8-byte loads in flight before the first use
    u32x2 rr[2][4];
    if (a.res) {
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          rr[n2][g] = nvalid[n2] ? *(const u32x2*)(a.res + poff[m] + n2 * nstep + 8 * g) : (u32x2){0u, 0u};
    }
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      if (!nvalid[n2]) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 k4 = *(const f32x4*)(kk + eoff[m] + n2 * 32 + 8 * g);
        float v0 = fmaxf(fmaf(rstd, acc[m][n2][4 * g + 0], k4.x), 0.f);
        float v1 = fmaxf(fmaf(rstd, acc[m][n2][4 * g + 1], k4.y), 0.f);
        float v2 = fmaxf(fmaf(rstd, acc[m][n2][4 * g + 2], k4.z), 0.f);
        float v3 = fmaxf(fmaf(rstd, acc[m][n2][4 * g + 3], k4.w), 0.f);
        if (a.res) {
          v0 += bf16_lo_to_f32(rr[n2][g].x); v1 += bf16_hi_to_f32(rr[n2][g].x);
          v2 += bf16_lo_to_f32(rr[n2][g].y); v3 += bf16_hi_to_f32(rr[n2][g].y);
        }
        s_sum += (v0 + v1) + (v2 + v3);
        s_sq = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, s_sq))));
        const u32x2 pk = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
        *(u32x2*)(a.y + poff[m] + n2 * nstep + 8 * g) = pk;
      }
    }
  }
  if (a.stats_out) {
    float* red = (float*)(smem + KK_OFF + KK_BYTES);
    s_sum = wave_sum(s_sum);
    s_sq = wave_sum(s_sq);
    if (lane == 0) { red[w] = s_sum; red[4 + w] = s_sq; }
    __syncthreads();
    if (tid == 0) {
      atomicAdd(a.stats_out + 2 * f, (double)((red[0] + red[1]) + (red[2] + red[3])));
      atomicAdd(a.stats_out + 2 * f + 1, (double)((red[4] + red[5]) + (red[6] + red[7])));
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
