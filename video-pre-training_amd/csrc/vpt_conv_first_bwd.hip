// Backward of the fused stack-0 first convolution (weights and bias only: its input is the uint8 image).
//
// Forward (vpt_conv_first.hip): P = maxpool3x3/2( relu( conv3x3(img/255, W) + b ) ); the 128x128xC pre-pool tensor
// is never stored, so this kernel RECOMPUTES it tile by tile exactly as the forward does (same MFMA path, same
// bf16 rounding), then for every pooled pixel and channel finds the arg-max conv pixel of its 3x3 window (first
// maximum in scan order, torch's rule; a zero maximum passes no gradient through the ReLU) and accumulates
//     dW[o][kh][kw][ch] += dP * img[argmax + (kh,kw)][ch] / 255 ,   db[o] += dP
// in registers: thread = (output channel, half of the tile's 64 pooled pixels), 28 fp32 accumulators, flushed with
// atomics once per workgroup.  Persistent workgroups (2 per CU) sweep the tile list so the flush is amortised.
// Replaces the autograd of lib/impala_cnn.py:86-97,115-117 for stack 0.
#include "vpt_common.h"
#include "vpt_kernels.h"

#define CT_RS 272
#define CT_BYTES (289 * CT_RS)
#define IN_OFF CT_BYTES
#define IN_BYTES 1088

__global__ __launch_bounds__(256, 2) void vpt_conv_first_bwd_kernel(VptConvFirstBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CT_BYTES + IN_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int tilesX = PW >> 3, tilesY = PH >> 3;
  const long T = (long)a.frames * tilesY * tilesX;
  const int nt = blockIdx.y;
  const int CB_out = a.Cout >> 5;

  op16x8 wfr[4][2];
#pragma unroll
  for (int cs = 0; cs < 4; ++cs)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wfr[cs][ks] = *((const op16x8*)a.wfrag + ((nt * 4 + cs) * 2 + ks) * 64 + lane);

  const int oc = tid & 127, half = tid >> 7;   // backward role: output channel within the N tile, pooled-pixel half
  const int og = nt * 128 + oc;
  const bool ovalid = og < a.Cout;
  float gw[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) gw[k] = 0.f;
  float gb = 0.f;
  unsigned char* in = smem + IN_OFF;

  // next tile's input bytes are fetched into registers while the current tile computes (as in the forward kernel)
  unsigned char nxt[5];
  auto fetch = [&](long tile) {
    long L = tile;
    const int tx = (int)(L % tilesX); L /= tilesX;
    const int ty = (int)(L % tilesY);
    const int f = (int)(L / tilesY);
    const int iy0 = 2 * (ty * 8) - 2, ix0 = 2 * (tx * 8) - 2;
    const uint8_t* img = a.img + (size_t)f * a.H * a.W * 3;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const int idx = tid + 256 * m;
      const int r = idx / 57, rem = idx - r * 57;
      const int y = iy0 + r, x = ix0 + rem / 3;
      const bool ok = idx < 19 * 57 && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const unsigned char v = img[ok ? ((long)y * a.W + ix0) * 3 + rem : 0];
      nxt[m] = ok ? v : (unsigned char)0;
    }
  };
  const long per = (T + gridDim.x - 1) / gridDim.x;
  const long t_begin = blockIdx.x * per, t_end = min(t_begin + per, T);
  if (t_begin < t_end) fetch(t_begin);

  for (long tile = t_begin; tile < t_end; ++tile) {
    long L = tile;
    const int tx = (int)(L % tilesX); L /= tilesX;
    const int ty = (int)(L % tilesY);
    const int f = (int)(L / tilesY);
    const int py0 = ty * 8, px0 = tx * 8;
#pragma unroll
    for (int m = 0; m < 5; ++m)
      if (tid + 256 * m < 19 * 57) in[tid + 256 * m] = nxt[m];
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);
    // this thread's 32 pooled gradients, two bf16 per register, fetched in four groups of 8: group 0 now (its latency
    // hides behind the recompute), group g + 1 while group g is routed
    const unsigned short* dP = (const unsigned short*)a.dpooled + ((size_t)(f * CB_out + ((ovalid ? og : 0) >> 5)) * PH * PW) * 32 + ((ovalid ? og : 0) & 31);
    auto load_group = [&](int g, uint32_t* dst) {
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        const int pp = half * 32 + g * 8 + q;     // pp and pp + 1 are neighbours in the same pooled row
        const size_t o = (size_t)((py0 + (pp >> 3)) * PW + px0 + (pp & 7)) * 32;
        dst[q >> 1] = (uint32_t)dP[o] | ((uint32_t)dP[o + 32] << 16);
      }
    };
    uint32_t cur[4], nx[4];
    load_group(0, cur);
    // ---- recompute the post-ReLU conv tile (identical to the forward kernel) ----
    for (int sub = w; sub < 10; sub += 4) {
      const int p = sub * 32 + l31;
      const bool pv = p < 289;
      const int pc = pv ? p : 288;
      const int cr = pc / 17, cc = pc - cr * 17;
      const int gy = 2 * py0 - 1 + cr, gx = 2 * px0 - 1 + cc;
      const bool inimg = pv && gy >= 0 && gx >= 0 && gy < a.H && gx < a.W;
      const unsigned char* ib = in + (cr * 19 + cc) * 3;
      op16x8 pf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float h[8];      // the byte as fp32: exact, so its bf16 is the upper half of the fp32 pattern
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kA = ks * 16 + e, kB = ks * 16 + 8 + e;     // k for lanes 0-31 / 32-63
        const int offA = kA + 48 * (kA / 9);                   // byte offset of tap (k/9, (k%9)/3), channel k%3
        const int offB = (kB < 27) ? (kB + 48 * (kB / 9)) : 0;
        float v = (float)ib[hi ? offB : offA];
        if (kB >= 27) v = hi ? ((kB <= 28) ? 1.0f : 0.0f) : v; // bias slots (k = 27, 28) carry 1.0, the rest 0
        h[e] = v;
      }
      u32x4 pk;
      pk.x = pack_op16x2_exact(h[0], h[1]);
      pk.y = pack_op16x2_exact(h[2], h[3]);
      pk.z = pack_op16x2_exact(h[4], h[5]);
      pk.w = pack_op16x2_exact(h[6], h[7]);
      pf[ks] = __builtin_bit_cast(op16x8, pk);
      }
      f32x16 acc[4];
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cs][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc[cs] = VPT_MFMA_32X32X16(wfr[cs][ks], pf[ks], acc[cs], 0, 0, 0);
      }
      // conv + bias (1/255 is folded into the weights), rounded to bf16 and stored RAW: the ReLU commutes with the
    // max-pool, so it is applied once per pooled value instead of once per conv value; pixels outside the image -> 0
    const uint32_t keep = inimg ? 0xffffffffu : 0u;
    if (pv) {
      unsigned char* dst = smem + p * CT_RS + hi * 8;
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 pk2 = {pack_op16x2(acc[cs][4 * g + 0], acc[cs][4 * g + 1]) & keep, pack_op16x2(acc[cs][4 * g + 2], acc[cs][4 * g + 3]) & keep};
          *(u32x2*)(dst + (cs * 32 + g * 8) * 2) = pk2;
        }
      }
    }
  }
  __syncthreads();
    // ---- arg-max routing + weight-gradient accumulation ----
    if (ovalid) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
      if (g < 3) load_group(g + 1, nx);
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) {
        const int pp = half * 32 + g * 8 + q8;
        const int pyl = pp >> 3, pxl = pp & 7;
        const float d = (q8 & 1) ? op16_hi_to_f32(cur[q8 >> 1]) : op16_lo_to_f32(cur[q8 >> 1]);
        if (d == 0.f) continue;
        const short* ct = (const short*)(smem + ((2 * pyl) * 17 + 2 * pxl) * CT_RS) + oc;
        short best = 0;                                        // raw bf16 patterns as signed integers: only values > 0 can win (ReLU gate)
        int bpos = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const short v = ct[(dy * 17 + dx) * (CT_RS / 2)];
            if (v > best) { best = v; bpos = dy * 17 + dx; }   // strict >: first maximum in scan order
          }
        if (best == 0) continue;                               // ReLU gate (and windows whose maximum is 0)
        const int cpos = (2 * pyl) * 17 + 2 * pxl + bpos;      // conv pixel within the 17x17 tile
        const int cr = cpos / 17, cc = cpos - cr * 17;
        const unsigned char* ib = in + (cr * 19 + cc) * 3;
        const float ds = d * (1.0f / 255.0f);                   // d(conv)/dW = img / 255
#pragma unroll
        for (int k = 0; k < 27; ++k) gw[k] = fmaf(ds, (float)ib[k + 48 * (k / 9)], gw[k]);
        gb += d;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = nx[k];
      __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  if (ovalid) {
#pragma unroll
    for (int k = 0; k < 27; ++k) atomicAdd(a.dw + (size_t)og * 27 + k, gw[k]);
    atomicAdd(a.db + og, gb);
  }
}

extern "C" int vpt_conv_first_bwd_launch(const VptConvFirstBwdArgs* a, hipStream_t stream) {
  if ((a->H & 15) || (a->W & 15) || (a->Cout & 31) || a->frames <= 0) return -1;
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                 ? prop.multiProcessorCount : 256;
  }
  const long tiles = (long)a->frames * (a->H >> 4) * (a->W >> 4);
  long gx = (long)num_cu * 2;
  if (tiles < gx) gx = tiles;
  hipLaunchKernelGGL(vpt_conv_first_bwd_kernel, dim3((unsigned)gx, (a->Cout + 127) / 128), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
