// Weight gradient of the 3x3 convolutions on bf16 MFMA (gfx950):
//     dW'[o][tap][c] = sum_{frames, pixels p} dacc[f][o][p] * x[f][c][p + tap]      (zero outside the image)
// i.e. the wgrad of FanInInitReLULayer's Conv2d (lib/util.py:58-65) with the GroupNorm gain already folded
// (vpt_conv_bwd_prep supplies dacc = rstd * dz; the host maps dW' to dW, dgain, dbias).
//
// The reduction runs over PIXELS, but both tensors are stored channel-fastest ([frame][C/32][H][W][32]), so a
// fragment's 8 consecutive k values (pixels) are 64 bytes apart in HBM.  They are transposed on the way into
// LDS: each 16-byte chunk (8 channels of one pixel) is scattered with eight ds_write_b16 into channel-major
// rows, after which both MFMA operands are plain ds_read_b128.  The +-1 pixel shifts of the three kernel
// columns would break the 16-byte alignment of those reads, so x is written three times, pre-shifted by
// 0/1/2 pixels (zero halo columns are cleared once and never overwritten).
//
// Workgroup = 128 couts (4 waves x 32) x 32 cins x 9 taps = 9 MFMA 32x32x16 accumulators per wave; one step
// = 64 pixels (64/W whole rows) of one frame; a workgroup sweeps `frames_per_wg` frames and adds its
// partial sums into the fp32 result with atomics (coalesced along cin).
#include "vpt_common.h"
#include "vpt_kernels.h"

#define WG_DT_RS 144                    // bytes per cout row of the transposed dacc tile (64 px + pad)
#define WG_DT_BYTES (128 * WG_DT_RS)    // 18432

__global__ __launch_bounds__(256, 2) void vpt_conv_wgrad_kernel(VptConvWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int W = a.W, H = a.H, HW = a.H * a.W;
  const int RB = 64 / W;                // rows per step
  const int Wp = W + 16;                // padded row pitch (elements) of the shifted x copies
  const int SC = (RB + 2) * Wp * 2 + 16;  // bytes per cin row (odd multiple of 16 -> conflict-free b128 reads)
  const int COPY = 32 * SC;
  unsigned char* DT = smem;
  unsigned char* XT = smem + WG_DT_BYTES;

  const int CBi = a.Cin >> 5, CBo = a.Cout >> 5;
  int L = blockIdx.x;
  const int cbi = L % CBi; L /= CBi;
  const int ot = L % a.OT;
  const int grp = L / a.OT;
  const int f0 = grp * a.frames_per_wg, f1 = min(f0 + a.frames_per_wg, a.frames);
  const int cbo = ot * 4 + w;           // this wave's 32 output channels
  const bool ovalid = cbo < CBo;

  // clear the x copies once (halo columns / pad stay zero for the whole kernel)
  for (int i = tid * 16; i < 3 * COPY; i += 256 * 16) *(u32x4*)(XT + i) = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int xchunks = (RB + 2) * W * 4;     // 16-byte chunks of the x slab per step (<= 768)
  const int steps_per_frame = H / RB;
  const int nsteps = (f1 - f0) * steps_per_frame;

  u32x4 dreg[4], xreg[3];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_step = [&](int s) {
    const int f = f0 + s / steps_per_frame, y0 = (s % steps_per_frame) * RB;
    if (ovalid) {
      const vpt_bf16* dp = a.dacc + ((size_t)(f * CBo + cbo) * HW + (size_t)y0 * W) * 32;
#pragma unroll
      for (int m = 0; m < 4; ++m) dreg[m] = *(const u32x4*)(dp + (lane + 64 * m) * 8);
    }
    const vpt_bf16* xp = a.x + ((size_t)(f * CBi + cbi) * HW) * 32;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int q = tid + 256 * m;
      u32x4 v = zero4;
      if (q < xchunks) {
        const int pix = q >> 2, part = q & 3;
        const int r = pix / W, x = pix - r * W;
        const int y = y0 - 1 + r;
        if (y >= 0 && y < H) v = *(const u32x4*)(xp + ((size_t)y * W + x) * 32 + part * 8);
      }
      xreg[m] = v;
    }
  };
  auto store_step = [&]() {
    if (ovalid) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int q = lane + 64 * m;          // chunk of this wave's 32-cout block: pixel q>>2, channels (q&3)*8..+7
        const int pix = q >> 2, part = q & 3;
        unsigned char* dst = DT + (w * 32 + part * 8) * WG_DT_RS + pix * 2;
        const uint32_t u[4] = {dreg[m].x, dreg[m].y, dreg[m].z, dreg[m].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *(unsigned short*)(dst + (2 * k) * WG_DT_RS) = (unsigned short)(u[k] & 0xffffu);
          *(unsigned short*)(dst + (2 * k + 1) * WG_DT_RS) = (unsigned short)(u[k] >> 16);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int q = tid + 256 * m;
      if (q < xchunks) {
        const int pix = q >> 2, part = q & 3;
        const int r = pix / W, x = pix - r * W;
        const uint32_t u[4] = {xreg[m].x, xreg[m].y, xreg[m].z, xreg[m].w};
#pragma unroll
        for (int s = 0; s < 3; ++s) {       // copy s holds image column j at index j + 9 - s
          unsigned char* dst = XT + s * COPY + (part * 8) * SC + (r * Wp + x + 9 - s) * 2;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            *(unsigned short*)(dst + (2 * k) * SC) = (unsigned short)(u[k] & 0xffffu);
            *(unsigned short*)(dst + (2 * k + 1) * SC) = (unsigned short)(u[k] >> 16);
          }
        }
      }
    }
  };

  if (nsteps > 0) load_step(0);
  for (int s = 0; s < nsteps; ++s) {
    store_step();
    __syncthreads();
    if (s + 1 < nsteps) load_step(s + 1);
    if (ovalid) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int q0 = ks * 16 + 8 * hi;       // first of this lane's 8 pixels within the 64-pixel step
        const int r = q0 / W, x0 = q0 - r * W;
        const bf16x8 af = *(const bf16x8*)(DT + (w * 32 + l31) * WG_DT_RS + q0 * 2);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const bf16x8 bfr = *(const bf16x8*)(XT + dx * COPY + l31 * SC + ((r + dy) * Wp + x0 + 8) * 2);
            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[dy * 3 + dx], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }

  if (ovalid) {
    const int c = cbi * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = cbo * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        atomicAdd(a.dw + ((size_t)o * 9 + t) * a.Cin + c, acc[t][r]);
      }
  }
}

extern "C" int vpt_conv_wgrad_launch(const VptConvWgradArgs* a_in, hipStream_t stream) {
  VptConvWgradArgs a = *a_in;
  if ((a.Cin & 31) || (a.Cout & 31) || a.frames <= 0 || (a.W != 16 && a.W != 32 && a.W != 64) || (a.H % (64 / a.W))) return -1;
  a.OT = (a.Cout + 127) / 128;
  const int tiles = a.OT * (a.Cin >> 5);
  int groups = (2048 + tiles - 1) / tiles;          // aim at ~2048 workgroups
  if (groups > a.frames) groups = a.frames;
  a.frames_per_wg = (a.frames + groups - 1) / groups;
  groups = (a.frames + a.frames_per_wg - 1) / a.frames_per_wg;
  const int RB = 64 / a.W, Wp = a.W + 16;
  const int SC = (RB + 2) * Wp * 2 + 16;
  const size_t lds = WG_DT_BYTES + 3 * 32 * SC;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vpt_conv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return -4;
    attr_set = true;
  }
  hipLaunchKernelGGL(vpt_conv_wgrad_kernel, dim3((unsigned)(tiles * groups)), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
