// Temporal Conv3d of the inverse dynamics model fused with the uint8 ingest, bias and ReLU (gfx950).
//
// Replaces ImgPreprocessing.forward (x/255, lib/policy.py:39-45) + InverseActionNet._conv3d_forward
// (lib/policy.py:394-403): Conv3d(3 -> O, kernel (5,1,1), padding (2,0,0)) over the T axis of each sequence
// + ReLU (FanInInitReLULayer without norm, so with bias; lib/policy.py:366-372).
//
// out[b,t,h,w,o] = relu( sum_{dt=-2..2} sum_c W[o,c,dt] * img[b,t+dt,h,w,c] / 255 + bias[o] ), zero outside [0,T).
// K = 15 (padded to 16): one MFMA 32x32x16 k-step with swapped operands (weights = A rows, pixels = B
// columns; 0..255 are exact in bf16, 1/255 applied in fp32), so a lane owns one pixel and groups of 4
// consecutive output channels -> 8-byte stores into the channel-blocked layout [frame][O/32][H][W][32].
// One workgroup = 256 consecutive pixels of one frame x 128 output channels; the five input frames' bytes
// (5 x 768 B) are staged in LDS with 16-byte loads.  Emits sum / sum-of-squares per frame for the GroupNorm
// of the following stack-0 firstconv (first_conv_norm=True in the IDM, lib/policy.py:360-363).
#include "vpt_common.h"
#include "vpt_kernels.h"

__global__ __launch_bounds__(256, 2) void vpt_conv3d_t5_kernel(VptConv3dArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char in[5 * 768];
  __shared__ __attribute__((aligned(16))) float bias_s[128];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int HW = a.H * a.W;
  const int chunks = HW >> 8;  // 256-pixel chunks per frame
  int L = blockIdx.x;
  const int nt = L % a.NT; L /= a.NT;
  const int chunk = L % chunks;
  const int f = L / chunks;       // frame index b*T + t
  const int t = f % a.T;
  const int p0 = chunk * 256;

  if (tid < 240) {
    const int dt = tid / 48, c16 = tid - dt * 48;  // 48 x 16 B = 768 B per frame slab
    const int tt = t + dt - 2;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (tt >= 0 && tt < a.T) v = *(const u32x4*)(a.img + ((size_t)(f + dt - 2) * HW + p0) * 3 + c16 * 16);
    *(u32x4*)(in + dt * 768 + c16 * 16) = v;
  }
  if (tid < 128) bias_s[tid] = a.bias[nt * 128 + tid];
  op16x8 wfr[4];
#pragma unroll
  for (int cs = 0; cs < 4; ++cs) wfr[cs] = *((const op16x8*)a.wfrag + (nt * 4 + cs) * 64 + lane);
  __syncthreads();

  const int CB_out = a.Cout >> 5;
  float s_sum = 0.f, s_sq = 0.f;
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int pl = w * 64 + sub * 32 + l31;  // pixel within the chunk
    float h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kA = e, kB = 8 + e;  // k for lanes 0-31 / 32-63; k = dt*3 + ch, k = 15 is padding
      const int offA = (kA / 3) * 768 + (kA % 3), offB = (kB < 15) ? (kB / 3) * 768 + (kB % 3) : 0;
      float v = (float)in[pl * 3 + (hi ? offB : offA)];   // a byte: exact in either 16-bit operand format
      if (kB >= 15) v = hi ? 0.f : v;
      h[e] = v;
    }
    const u32x4 pk = {pack_op16x2_exact(h[0], h[1]), pack_op16x2_exact(h[2], h[3]), pack_op16x2_exact(h[4], h[5]), pack_op16x2_exact(h[6], h[7])};
    const op16x8 pf = __builtin_bit_cast(op16x8, pk);
    const size_t pbase = ((size_t)f * CB_out + nt * 4) * HW * 32 + (size_t)(p0 + pl) * 32 + 4 * hi;
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      if (nt * 4 + cs >= CB_out) continue;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = VPT_MFMA_32X32X16(wfr[cs], pf, acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *(const f32x4*)(bias_s + cs * 32 + 8 * g + 4 * hi);
        const float v0 = fmaxf(fmaf(acc[4 * g + 0], 1.0f / 255.0f, b4.x), 0.f);
        const float v1 = fmaxf(fmaf(acc[4 * g + 1], 1.0f / 255.0f, b4.y), 0.f);
        const float v2 = fmaxf(fmaf(acc[4 * g + 2], 1.0f / 255.0f, b4.z), 0.f);
        const float v3 = fmaxf(fmaf(acc[4 * g + 3], 1.0f / 255.0f, b4.w), 0.f);
        s_sum += (v0 + v1) + (v2 + v3);
        s_sq = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, s_sq))));
        const u32x2 o = {pack_op16x2(v0, v1), pack_op16x2(v2, v3)};
        *(u32x2*)(a.y + pbase + (size_t)cs * HW * 32 + 8 * g) = o;
      }
    }
  }
  if (a.stats_out) {
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

extern "C" int vpt_conv3d_launch(const VptConv3dArgs* a, hipStream_t stream) {
  if (((a->H * a->W) & 255) || (a->Cout & 31) || a->frames <= 0 || a->T <= 0 || (a->frames % a->T)) return -1;
  const long grid = (long)a->frames * ((a->H * a->W) >> 8) * a->NT;
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_conv3d_t5_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
