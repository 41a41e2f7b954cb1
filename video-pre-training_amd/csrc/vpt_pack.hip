// Weight re-packing on the device: the reference's fp32 tensors (the `.weights` state_dict) -> the layouts
// the kernels stream, so that a host without torch (or packing.py) can drive the library, and so that the re-pack after
// every optimiser step is two launches per layer instead of a dozen tensor ops.
//
//  vpt_pack_conv3x3_kernel : Conv2d weight [Cout][Cin][3][3] with the preceding GroupNorm(1, Cin) affine folded
//                            (lib/util.py:58-82, lib/impala_cnn.py:30-52) -> wpk [NT][Cin/32][9][128][32] (16-bit,
//                            = round(W * gain), 16-byte chunks of each 64-byte row XOR-swizzled by ((cout >> 2) & 3): the LDS
//                            image vpt_conv3x3_kernel DMAs) + the edge tables SA / SG [9][NT*128] fp32 of its GroupNorm fold.
//  vpt_pack_linear_kernel  : nn.Linear weight [N][K] -> [ceil(N/128)][K/32][128][32] (16-bit, rows >= N zero).
//  vpt_pack_conv_first_kernel : stack-0 firstconv weight [Cout][3][3][3] + bias -> the MFMA A-operand fragments of
//                            vpt_conv_first_kernel ([NT][4][2][64][8]: W / 255 at k = (kh*3+kw)*3+ch, the bias as hi / lo halves at k = 27 / 28; the 32 K
//                            slots in the order of vpt_conv_first_tile.h: per kernel row eight contiguous values, the ninth ones + bias last).
//  vpt_pack_conv3d_t5_kernel : IDM Conv3d weight [O][3][5][1][1] -> fragments [NT][4][64][8] (k = dt*3+ch, k = 15 zero) + padded bias.
//  vpt_chw_to_blocked_kernel : fp32 [rows][C*H*W] in the reference's C,H,W flatten order (lib/impala_cnn.py:192-193) -> the blocked
//                            activation order [rows][C/32][H][W][32] (dense-layer weight columns, its LayerNorm gain / bias).
// All reproduce video-pre-training_amd/packing.py bit for bit (tests/test_gpu_kernels.py): sums in fp64, round to nearest even.
#include "vpt_common.h"
#include "vpt_kernels.h"

__global__ __launch_bounds__(256) void vpt_pack_conv3x3_kernel(VptPackConvArgs a) {
  __shared__ double red[2][9][4];
  const int o = blockIdx.x;            // padded cout index, 0 .. NT*128-1
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nt = o >> 7, r = o & 127;
  const int n = a.Cin * 9;
  double sg[9], sa[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { sg[t] = 0.0; sa[t] = 0.0; }
  for (int i = tid; i < n; i += 256) {
    const int c = i / 9, t = i - c * 9;
    float wv = 0.f, wgv = 0.f;
    if (o < a.Cout) {
      wv = a.weight[(size_t)o * n + i];
      wgv = wv * a.gain[c];
    }
    asm volatile("" : "+v"(wgv));          // keep the fp32 product a value of its own: a fused multiply-convert (v_fma_mix)
    const op16_t h = (op16_t)wgv;          // would round once instead of twice and differ from packing.py in rare cases
    // [nt][cb][tap][row r][32 cin]: chunk (c & 31) >> 3 of the row stored at chunk ^ ((r >> 2) & 3)
    const int cb = c >> 5, ci = c & 31;
    const int chunk = (ci >> 3) ^ ((r >> 2) & 3);
    a.wpk[((((size_t)nt * (a.Cin >> 5) + cb) * 9 + t) * 128 + r) * 32 + chunk * 8 + (ci & 7)] = h;
    if (a.edge_sa) {
      const double hg = (double)(float)h, wb = (double)wv * (double)a.bias[c];
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) { if (tt == t) { sg[tt] += hg; sa[tt] += wb; } }
    }
  }
  if (!a.edge_sa) return;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    double x = sg[t], y = sa[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { x += __shfl_xor(x, off, 64); y += __shfl_xor(y, off, 64); }
    if (lane == 0) { red[0][t][w] = x; red[1][t][w] = y; }
  }
  __syncthreads();
  if (tid < 18) {
    const int which = tid / 9, e = tid % 9;       // edge class e = 3 * ey + ex: taps inside the image for that class
    const int ey = e / 3, ex = e % 3;
    double s = 0.0;
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        const bool vy = (ey == 0) ? (kh >= 1) : (ey == 2 ? (kh <= 1) : true);
        const bool vx = (ex == 0) ? (kw >= 1) : (ex == 2 ? (kw <= 1) : true);
        if (vy && vx) { const int t = kh * 3 + kw; s += (red[which][t][0] + red[which][t][1]) + (red[which][t][2] + red[which][t][3]); }
      }
    (which ? a.edge_sa : a.edge_sg)[(size_t)e * (a.NT * 128) + o] = (float)s;
  }
}

extern "C" int vpt_pack_conv3x3_launch(const VptPackConvArgs* a, hipStream_t stream) {
  if ((a->Cin & 31) || (a->Cout & 31) || a->Cout <= 0) return -1;
  hipLaunchKernelGGL(vpt_pack_conv3x3_kernel, dim3(a->NT * 128), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

__global__ __launch_bounds__(256) void vpt_pack_linear_kernel(const float* __restrict__ w, op16_t* __restrict__ out, int N, int K, int transposed, int ldw, int src_rows) {
  // out[nt][kb][r][32]: element (n = nt*128 + r, k = kb*32 + j) = W[n][k], or, `transposed`, W[k][n] of a [src_rows][ldw] source
  // (k >= src_rows -> 0: the reduction dimension padded to a multiple of 64)
  const size_t total = (size_t)((N + 127) / 128) * 128 * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int j = (int)(i & 31);
    size_t q = i >> 5;
    const int r = (int)(q & 127); q >>= 7;
    const int kb = (int)(q % (size_t)(K >> 5));
    const int nt = (int)(q / (size_t)(K >> 5));
    const int n = nt * 128 + r, k = kb * 32 + j;
    float v = 0.f;
    if (n < N) v = transposed ? (k < src_rows ? w[(size_t)k * ldw + n] : 0.f) : w[(size_t)n * ldw + k];
    out[i] = (op16_t)v;
  }
}

extern "C" int vpt_pack_linear_launch(const float* w, void* out, int N, int K, int transposed, int ldw, int src_rows, hipStream_t stream) {
  if (N <= 0 || K <= 0 || (K & 63)) return -1;
  const size_t total = (size_t)((N + 127) / 128) * 128 * K;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(vpt_pack_linear_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, (op16_t*)out, N, K, transposed, ldw, src_rows);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- stack-0 first conv: W[o][ch][kh][kw], bias[o] -> frag[nt][cs][ks][hi][l31][8]  (o = nt*128 + cs*32 + l31, slot = ks*16 + hi*8 + e;
// slots 0..23 hold k = 9 * (slot / 8) + slot % 8 (values 0..7 of kernel row 0, 1, 2), slots 24..26 k = 8, 17, 26, slots 27.. k = slot)
__global__ __launch_bounds__(256) void vpt_pack_conv_first_kernel(const float* __restrict__ w, const float* __restrict__ bias, op16_t* __restrict__ out, int Cout, int NT) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NT * 128 * 32) return;
  const int e = i & 7, l31 = (i >> 3) & 31, hi = (i >> 8) & 1, ks = (i >> 9) & 1, cs = (i >> 10) & 3, nt = i >> 12;
  const int o = nt * 128 + cs * 32 + l31, slot = ks * 16 + hi * 8 + e;
  const int k = (slot < 24) ? 9 * (slot >> 3) + (slot & 7) : (slot < 27) ? 9 * (slot - 24) + 8 : slot;
  float v = 0.f;
  if (o < Cout) {
    if (k < 27) {
      const int tap = k / 3, ch = k - 3 * tap;              // tap = kh*3 + kw
      v = __fdiv_rn(w[(size_t)o * 27 + ch * 9 + tap], 255.0f);   // correctly rounded fp32 quotient, as torch's W / 255 (the pixel operand is the raw byte)
    } else if (k == 27 || k == 28) {
      const float b = bias[o];
      const float bh = (float)(op16_t)b;
      v = (k == 27) ? bh : (b - bh);                         // hi / lo halves: the bias enters the fp32 accumulator to ~16 mantissa bits
    }
  }
  asm volatile("" : "+v"(v));                                // the quotient / difference is an fp32 value of its own before the 16-bit rounding
  out[i] = (op16_t)v;
}

extern "C" int vpt_pack_conv_first_launch(const float* w, const float* bias, void* out, int Cout, hipStream_t stream) {
  if (Cout <= 0 || (Cout & 31)) return -1;
  const int NT = (Cout + 127) / 128;
  hipLaunchKernelGGL(vpt_pack_conv_first_kernel, dim3(NT * 16), dim3(256), 0, stream, w, bias, (op16_t*)out, Cout, NT);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- IDM temporal conv: W[o][ch][dt], bias[o] -> frag[nt][cs][hi][l31][8] (k = hi*8 + e = dt*3 + ch; k = 15 zero), bias_pad[NT*128]
__global__ __launch_bounds__(256) void vpt_pack_conv3d_t5_kernel(const float* __restrict__ w, const float* __restrict__ bias, op16_t* __restrict__ out,
                                                                float* __restrict__ bias_pad, int O, int NT) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NT * 128 * 16) return;
  const int e = i & 7, l31 = (i >> 3) & 31, hi = (i >> 8) & 1, cs = (i >> 9) & 3, nt = i >> 11;
  const int o = nt * 128 + cs * 32 + l31, k = hi * 8 + e;
  float v = 0.f;
  if (o < O && k < 15) {
    const int dt = k / 3, ch = k - 3 * dt;
    v = w[(size_t)o * 15 + ch * 5 + dt];
  }
  out[i] = (op16_t)v;
  if (k == 0) bias_pad[o] = (o < O) ? bias[o] : 0.f;
}

extern "C" int vpt_pack_conv3d_t5_launch(const float* w, const float* bias, void* out, float* bias_pad, int O, hipStream_t stream) {
  if (O <= 0 || (O & 31)) return -1;
  const int NT = (O + 127) / 128;
  hipLaunchKernelGGL(vpt_pack_conv3d_t5_kernel, dim3(NT * 8), dim3(256), 0, stream, w, bias, (op16_t*)out, bias_pad, O, NT);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- C,H,W flatten order -> blocked activation order, per row: dst[r][cb][h][w][j] = src[r][(cb*32 + j)*H*W + h*W + w]
__global__ __launch_bounds__(256) void vpt_chw_to_blocked_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int C, int HW) {
  const size_t per = (size_t)C * HW, total = (size_t)rows * per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / per, q = i - r * per;
    const int j = (int)(q & 31);
    const size_t p = q >> 5;                 // cb * HW + (h*W + w)
    const int cb = (int)(p / HW), hw = (int)(p - (size_t)cb * HW);
    dst[i] = src[r * per + (size_t)(cb * 32 + j) * HW + hw];
  }
}

extern "C" int vpt_chw_to_blocked_launch(const float* src, float* dst, long rows, int C, int H, int W, hipStream_t stream) {
  if (rows <= 0 || C <= 0 || (C & 31) || H <= 0 || W <= 0) return -1;
  const size_t total = (size_t)rows * C * H * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(vpt_chw_to_blocked_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, rows, C, H * W);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
