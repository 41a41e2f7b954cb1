// Weight gradient of the 3x3 convolutions on bf16 MFMA (gfx950):
//     dW'[o][tap][c] = sum_{frames, pixels p} dacc[f][o][p] * x[f][c][p + tap]      (zero outside the image)
// i.e. the wgrad of FanInInitReLULayer's Conv2d (lib/util.py:58-65) with the GroupNorm gain already folded
// (vpt_conv_bwd_prep supplies dacc = rstd * dz; the host maps dW' to dW, dgain, dbias).
//
// The reduction runs over PIXELS while both tensors are stored channel-fastest ([frame][C/32][H][W][32]), so
// an MFMA fragment's 8 consecutive k values (pixels) are 64 bytes apart.  gfx950's LDS transpose read
// (ds_read_b64_tr_b16: 16 lanes read a [4 pixel][16 channel] block and each receives one channel's 4 pixels)
// does that transposition for free, so LDS holds both tiles exactly as they lie in HBM (plain 16-byte copies).
//
// Operand reuse: with k = the x pixel q of row r+dy, tap (dy,dx) is  sum_q dacc[o][r][q-dx+1] x[c][r+dy][q]:
// the three kernel COLUMNS are three 64-byte-shifted reads of the dacc row (zero halo columns in LDS), the three
// kernel ROWS three x rows -- 3 + 3 fragment reads feed 9 MFMAs.
//
// Workgroup = 8 waves = 4 (32-cout blocks of a 128-cout tile) x 2 (32-cin blocks); a wave owns the 9 taps of its
// (cout block, cin block) pair = 9 MFMA 32x32x16 accumulators.  One step = 64 pixels (64/W image rows) of one
// frame, double-buffered in LDS with the next step's global loads in flight during the MFMAs.  A workgroup
// sweeps a contiguous range of frames (the grid is one workgroup per CU) and writes its partial sums to scratch;
// vpt_conv_wgrad_reduce_kernel adds the per-group partials into the fp32 result.
#include "vpt_common.h"
#include "vpt_kernels.h"

__device__ __forceinline__ op16x8 tr_frag(const unsigned char* p) {  // 8 pixels of this lane's channel: two 4-pixel transposes
  const op16x4 a = lds_tr16_read(p);
  const op16x4 b = lds_tr16_read(p + 256);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

#ifndef VPT_WGRAD_PXS
#define VPT_WGRAD_PXS 128               // pixels of one frame per step (one barrier per step)
#endif

template <int W>
__global__ __launch_bounds__(512, 1) void vpt_conv_wgrad_kernel(VptConvWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PXS = VPT_WGRAD_PXS;
  constexpr int RB = PXS / W;           // image rows per step
  constexpr int XR = RB + 2;            // x rows per step (halo rows above / below)
  constexpr int DP = W + 2;             // dacc pixels per row in LDS (halo columns, kept zero)
  constexpr int DCB = RB * DP * 64;     // bytes of one cout block's dacc slab
  constexpr int XCB = XR * W * 64;      // bytes of one cin block's x slab
  constexpr int BUF = 4 * DCB + 2 * XCB;
  constexpr int NXC = 2 * XR * W * 4;   // 16-byte x chunks per step
  constexpr int NX = (NXC + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = w & 3, wi = w >> 2;
  const int hi = lane >> 5, l31 = lane & 31;
  const int H = a.H, HW = a.H * W;
  const int CBi = a.Cin >> 5, CBo = a.Cout >> 5;
  const int CP = (CBi + 1) >> 1;
  const int L = blockIdx.x;
  const int tiles = a.OT * CP;
  const int tile = L % tiles, grp = L / tiles;
  const int cp = tile % CP, ot = tile / CP;
  const int f0 = grp * a.frames_per_wg, f1 = min(f0 + a.frames_per_wg, a.frames);

  // dacc halo columns of both buffers: zero once, never overwritten
  for (int i = tid; i < 2 * 4 * RB * 2 * 4; i += 512) {
    int q = i;
    const int part = q & 3; q >>= 2;
    const int side = q & 1; q >>= 1;
    const int r = q % RB; q /= RB;
    const int cb = q & 3, buf = q >> 2;
    *(u32x4*)(smem + buf * BUF + cb * DCB + (r * DP + side * (W + 1)) * 64 + part * 16) = (u32x4){0u, 0u, 0u, 0u};
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int steps_per_frame = H / RB;
  const int nsteps = (f1 - f0) * steps_per_frame;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  constexpr int ND = 4 * PXS * 4 / 512;
  u32x4 dreg[ND], xreg[NX];

  // Staging addresses = wave-uniform base of the step (frame, first image row: scalar arithmetic) + a per-lane byte offset computed
  // once.  Interior steps of fully valid tiles -- no image row above / below missing, no channel block beyond Cin / Cout --
  // load straight through; the first version recomputed every 64-bit address per lane and step (18 quarter-rate integer
  // multiplies + 30 selects per step, ~25 % of the step's MFMA time in vector issue slots).
  unsigned doff[ND], xoff[NX];
  unsigned xfirst = 0, xlast = 0;   // bit m: chunk m belongs to the halo row above / below the step's rows
#pragma unroll
  for (int m = 0; m < ND; ++m) {
    const int q = tid + 512 * m;
    const int cbo = min(ot * 4 + q / (PXS * 4), CBo - 1);
    doff[m] = (unsigned)(cbo * HW * 32 + (q % (PXS * 4)) * 8) * 2u;
  }
#pragma unroll
  for (int m = 0; m < NX; ++m) {
    const int q = min(tid + 512 * m, NXC - 1);
    const int cb = q / (XR * W * 4), rem = q % (XR * W * 4);
    const int cbi = min(cp * 2 + cb, CBi - 1);
    xoff[m] = (unsigned)((cbi * HW) * 32 + rem * 8) * 2u;   // rem = (row r, pixel, part): rows are W * 64 bytes apart in HBM too
    const int r = rem / (W * 4);
    xfirst |= (r == 0) ? (1u << m) : 0u;
    xlast |= (r == XR - 1) ? (1u << m) : 0u;
  }
  const bool all_valid = (ot * 4 + 4 <= CBo) && (cp * 2 + 2 <= CBi);
  auto load_step = [&](int s) {
    const int f = f0 + s / steps_per_frame, y0 = (s % steps_per_frame) * RB;
    if (all_valid && y0 > 0 && y0 + RB < H) {
      const char* dbase = (const char*)(a.dacc + ((size_t)f * CBo * HW + (size_t)y0 * W) * 32);
      const char* xbase = (const char*)(a.x + ((size_t)f * CBi * HW + (size_t)(y0 - 1) * W) * 32);
#pragma unroll
      for (int m = 0; m < ND; ++m) dreg[m] = *(const u32x4*)(dbase + doff[m]);
#pragma unroll
      for (int m = 0; m < NX; ++m) xreg[m] = *(const u32x4*)(xbase + xoff[m]);
      return;
    }
    if (all_valid) {   // first / last rows of a frame: the halo row outside the image is fetched from its neighbour row and zeroed
      const bool top = y0 == 0, bot = y0 + RB >= H;
      const char* dbase = (const char*)(a.dacc + ((size_t)f * CBo * HW + (size_t)y0 * W) * 32);
      const char* xbase = (const char*)(a.x + ((size_t)f * CBi * HW) * 32) + ((long)(y0 - 1) * W) * 64;
#pragma unroll
      for (int m = 0; m < ND; ++m) dreg[m] = *(const u32x4*)(dbase + doff[m]);
#pragma unroll
      for (int m = 0; m < NX; ++m) {
        const bool zt = top && ((xfirst >> m) & 1u), zb = bot && ((xlast >> m) & 1u);
        const int adj = zt ? W * 64 : (zb ? -W * 64 : 0);
        const u32x4 v = *(const u32x4*)(xbase + (long)xoff[m] + adj);
        xreg[m] = (zt || zb) ? zero4 : v;
      }
      return;
    }
    // partial tiles (Cout not a multiple of 128 / Cin of 64): every load executes from a clamped (valid) address and is zeroed
    // afterwards when it lies outside the image / beyond the channel count
#pragma unroll
    for (int m = 0; m < ND; ++m) {         // 4 cout blocks x PXS pixels x 4 parts chunks; a block's PXS pixels are contiguous in HBM
      const int q = tid + 512 * m;
      const int cbo = ot * 4 + q / (PXS * 4);
      const u32x4 v = *(const u32x4*)(a.dacc + ((size_t)(f * CBo + min(cbo, CBo - 1)) * HW + (size_t)y0 * W) * 32 + (q % (PXS * 4)) * 8);
      dreg[m] = (cbo < CBo) ? v : zero4;
    }
#pragma unroll
    for (int m = 0; m < NX; ++m) {
      const int q = min(tid + 512 * m, NXC - 1);
      const int cb = q / (XR * W * 4), rem = q % (XR * W * 4);
      const int r = rem / (W * 4), rem2 = rem % (W * 4);
      const int y = y0 - 1 + r, cbi = cp * 2 + cb;
      const u32x4 v = *(const u32x4*)(a.x + ((size_t)(f * CBi + min(cbi, CBi - 1)) * HW + (size_t)min(max(y, 0), H - 1) * W) * 32 + rem2 * 8);
      xreg[m] = (y >= 0 && y < H && cbi < CBi) ? v : zero4;
    }
  };
  auto store_step = [&](int buf) {
    unsigned char* base = smem + buf * BUF;
#pragma unroll
    for (int m = 0; m < ND; ++m) {
      const int q = tid + 512 * m;
      const int pix = (q % (PXS * 4)) >> 2, part = q & 3;
      *(u32x4*)(base + (q / (PXS * 4)) * DCB + ((pix / W) * DP + (pix % W) + 1) * 64 + part * 16) = dreg[m];
    }
#pragma unroll
    for (int m = 0; m < NX; ++m) {
      const int q = tid + 512 * m;
      if (NXC % 512 == 0 || q < NXC) *(u32x4*)(base + 4 * DCB + q * 16) = xreg[m];   // x slabs are stored exactly in chunk order
    }
  };

  // per-lane part of every fragment address: 16-lane group g reads pixels 8*(g>>1) + (i>>2) (+4), channels 16*(g&1) + 4*(i&3)..+3
  const int g16 = lane >> 4, i16 = lane & 15;
  const int lane_off = (8 * (g16 >> 1) + (i16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;

  if (nsteps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    load_step(min(s + 1, nsteps - 1));   // the last iteration re-stages its own step into the idle buffer (harmless)
    const unsigned char* base = smem + (s & 1) * BUF;
    const unsigned char* dA = base + wo * DCB + lane_off;
    const unsigned char* xB = base + 4 * DCB + wi * XCB + lane_off;
    // Two fragment register sets: while the nine MFMAs of 16-pixel slice ks issue, the six fragments of slice ks + 1 are
    // requested, one behind each of the first six MFMAs (pinned with sched_barrier: left alone the scheduler sinks every
    // read next to its use and the matrix pipe drains for an LDS round trip per slice -- the same finding as in
    // vpt_conv3x3_kernel, round 2).  Only slice 0 of a step, whose buffer was filled behind the barrier, is exposed.
    op16x8 af[2][3], bfr[2][3];
#define WG_SB() __builtin_amdgcn_sched_barrier(0)
#define WG_LDA(set_, ks_, dx_) af[set_][dx_] = tr_frag(dA + ((((ks_) * 16) / W) * DP + (((ks_) * 16) % W) - (dx_) + 2) * 64)
#define WG_LDB(set_, ks_, dy_) bfr[set_][dy_] = tr_frag(xB + (((((ks_) * 16) / W) + (dy_)) * W + (((ks_) * 16) % W)) * 64)
#define WG_MM(set_, t_) acc[t_] = VPT_MFMA_32X32X16(af[set_][(t_) % 3], bfr[set_][(t_) / 3], acc[t_], 0, 0, 0)
#define WG_SLICE(set_, ks_, NEXT)                                                                         \
  do {                                                                                                    \
    WG_MM(set_, 0); if (NEXT) WG_LDA(1 - (set_), (ks_) + 1, 0); WG_SB();                                  \
    WG_MM(set_, 1); if (NEXT) WG_LDB(1 - (set_), (ks_) + 1, 0); WG_SB();                                  \
    WG_MM(set_, 2); if (NEXT) WG_LDA(1 - (set_), (ks_) + 1, 1); WG_SB();                                  \
    WG_MM(set_, 3); if (NEXT) WG_LDA(1 - (set_), (ks_) + 1, 2); WG_SB();                                  \
    WG_MM(set_, 4); if (NEXT) WG_LDB(1 - (set_), (ks_) + 1, 1); WG_SB();                                  \
    WG_MM(set_, 5); if (NEXT) WG_LDB(1 - (set_), (ks_) + 1, 2); WG_SB();                                  \
    WG_MM(set_, 6); WG_MM(set_, 7); WG_MM(set_, 8); WG_SB();                                              \
  } while (0)
    WG_LDA(0, 0, 0); WG_LDB(0, 0, 0); WG_LDA(0, 0, 1); WG_LDA(0, 0, 2); WG_LDB(0, 0, 1); WG_LDB(0, 0, 2);
    WG_SB();
    WG_SLICE(0, 0, true);
    WG_SLICE(1, 1, true);
    WG_SLICE(0, 2, true);
    if constexpr (PXS == 64) {
      WG_SLICE(1, 3, false);
    } else {
      WG_SLICE(1, 3, true);
      WG_SLICE(0, 4, true);
      WG_SLICE(1, 5, true);
      WG_SLICE(0, 6, true);
      WG_SLICE(1, 7, false);
    }
#undef WG_SLICE
#undef WG_MM
#undef WG_LDA
#undef WG_LDB
#undef WG_SB
    store_step((s + 1) & 1);
    __syncthreads();
  }

  // partial sums of this frame group -> scratch [grp][Cout][9][Cin]
  const int cbi = cp * 2 + wi, cbo = ot * 4 + wo;
  if (cbi < CBi && cbo < CBo) {
    float* part = a.partial + (size_t)grp * a.Cout * 9 * a.Cin;
    // element (o, t, c) with o = cbo * 32 + 4 hi + (r & 3) + 8 (r >> 2): one per-lane base, the rest are uniform multiples of Cin
    float* pl = part + ((size_t)(cbo * 32 + 4 * hi) * 9) * a.Cin + cbi * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) pl[(size_t)((((r & 3) + 8 * (r >> 2)) * 9 + t)) * a.Cin] = acc[t][r];
  }
}

// dw[i] += sum over frame groups of partial[g][i]
__global__ __launch_bounds__(256) void vpt_conv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int n4, int groups) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < groups; ++g) s += *((const f32x4*)partial + (size_t)g * n4 + i);
  f32x4* d = (f32x4*)dw + i;
  *d = *d + s;
}

extern "C" int vpt_conv_wgrad_groups(int frames, int Cin, int Cout) {
  const int tiles = ((Cout + 127) / 128) * (((Cin >> 5) + 1) >> 1);
  int groups = 256 / tiles;
  if (groups < 1) groups = 1;
  if (groups > frames) groups = frames;
  const int fpg = (frames + groups - 1) / groups;
  return (frames + fpg - 1) / fpg;
}

extern "C" int vpt_conv_wgrad_launch(const VptConvWgradArgs* a_in, hipStream_t stream) {
  VptConvWgradArgs a = *a_in;
  if ((a.Cin & 31) || (a.Cout & 31) || a.frames <= 0 || (a.W != 16 && a.W != 32 && a.W != 64) || (a.H % (VPT_WGRAD_PXS / a.W)) || !a.partial) return -1;
  a.OT = (a.Cout + 127) / 128;
  const int tiles = a.OT * (((a.Cin >> 5) + 1) >> 1);
  const int groups = vpt_conv_wgrad_groups(a.frames, a.Cin, a.Cout);
  a.frames_per_wg = (a.frames + groups - 1) / groups;
  const int RB = VPT_WGRAD_PXS / a.W;
  const size_t lds = 2 * (size_t)(4 * RB * (a.W + 2) * 64 + 2 * (RB + 2) * a.W * 64);
  static unsigned long long optin_done[3] = {0, 0, 0};
  if (!vpt_lds_optin((const void*)vpt_conv_wgrad_kernel<64>, 160 * 1024, &optin_done[0]) ||
      !vpt_lds_optin((const void*)vpt_conv_wgrad_kernel<32>, 160 * 1024, &optin_done[1]) ||
      !vpt_lds_optin((const void*)vpt_conv_wgrad_kernel<16>, 160 * 1024, &optin_done[2]))
    return -4;
  const dim3 grid((unsigned)(tiles * groups));
  if (a.W == 64) hipLaunchKernelGGL(vpt_conv_wgrad_kernel<64>, grid, dim3(512), lds, stream, a);
  else if (a.W == 32) hipLaunchKernelGGL(vpt_conv_wgrad_kernel<32>, grid, dim3(512), lds, stream, a);
  else hipLaunchKernelGGL(vpt_conv_wgrad_kernel<16>, grid, dim3(512), lds, stream, a);
  const int n4 = a.Cout * 9 * a.Cin / 4;
  hipLaunchKernelGGL(vpt_conv_wgrad_reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, stream, a.partial, a.dw, n4, groups);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
