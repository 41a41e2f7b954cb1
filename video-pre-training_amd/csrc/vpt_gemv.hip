// Skinny linear layer for the acting path (M = B*T <= 8 rows, agent.py:190-206): out[m][n] = sum_k A[m][k] W[n][k]
// with the same fused epilogue as vpt_gemm_kernel (bias, ReLU, gate mask, fp32 residual, fp32 / bf16 outputs, or
// split-K partial slices).  At M = 1 the layer is a stream of the weight matrix (2 bytes per MAC): HBM-bound, so the job is
// to keep many 16-byte loads in flight on every CU rather than to feed the MFMA -- a 256 x 128 GEMM tile would
// leave 1 row of 256 busy and put N/128 workgroups on a 256-CU chip.
//
// Workgroup = 16 output columns (rows of W) of one packed 128-row tile; the packed layout [NT][K/32][128][32] makes
// those 16 rows x 32 k one contiguous 1 KB run = one 16-byte load per lane of a wave; the 4 waves interleave over
// the k blocks.  Lane = (row, 8-k chunk); fp32 accumulate; reduce over the 4 chunk lanes by shuffles and over the
// waves through LDS.  grid = ceil(N/16) x splitk.
#include "vpt_common.h"
#include "vpt_kernels.h"

template <int MR>
__global__ __launch_bounds__(256) void vpt_gemv_kernel(VptGemmArgs a) {
  __shared__ float part_[4][16][MR];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int NB = (a.N + 15) >> 4;
  const int nb = blockIdx.x % NB, split = blockIdx.x / NB;
  const int r0 = nb * 16, nt = r0 >> 7, rin = r0 & 127;
  const int row = lane >> 2, chunk = lane & 3;
  const int kbs = a.K >> 5;
  const int per = (kbs + a.splitk - 1) / a.splitk;
  const int kb0 = split * per, kb1 = min(kb0 + per, kbs);
  const vpt_op16* wp = a.wpk + ((size_t)nt * kbs * 128 + rin + row) * 32 + chunk * 8;
  const vpt_op16* ap = a.A + chunk * 8;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
#pragma unroll 4
  for (int kb = kb0 + w; kb < kb1; kb += 4) {
    float wv[8];
    unpack8(*(const u32x4*)(wp + (size_t)kb * 4096), wv);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float av[8];
      unpack8(*(const u32x4*)(ap + (size_t)min(m, a.M - 1) * a.lda + kb * 32), av);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[m] = fmaf(wv[k], av[k], acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    acc[m] += __shfl_xor(acc[m], 1, 64);
    acc[m] += __shfl_xor(acc[m], 2, 64);
  }
  if (chunk == 0) {
#pragma unroll
    for (int m = 0; m < MR; ++m) part_[w][row][m] = acc[m];
  }
  __syncthreads();
  if (tid < 16 * MR) {
    const int m = tid / 16, r = tid % 16;     // consecutive threads -> consecutive output columns
    const int col = r0 + r;
    if (m < a.M && col < a.N) {
      float v = (part_[0][r][m] + part_[1][r][m]) + (part_[2][r][m] + part_[3][r][m]);
      if (a.bias && split == 0) v += a.bias[col];
      if (a.atomic_out) {   // split-K: own [M][ldc] slice per split, summed by the caller (deterministic)
        a.out_f32[((size_t)split * a.M + m) * a.ldc + col] = v;
      } else {
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.mask && !((float)a.mask[(size_t)m * a.ldm + col] > 0.f)) v = 0.f;
        if (a.res) v += a.res[(size_t)m * a.ldr + col];
        if (a.out_f32) a.out_f32[(size_t)m * a.ldc + col] = v;
        if (a.out_bf16) a.out_bf16[(size_t)m * a.ldcb + col] = (vpt_op16)v;
      }
    }
  }
}

extern "C" int vpt_gemv_launch(const VptGemmArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->M > 8 || a->N <= 0 || (a->K & 31) || a->splitk < 1 || (a->lda & 7)) return -1;
  if (a->splitk > 1 && (!a->atomic_out || a->relu || a->res || a->out_bf16 || a->mask)) return -1;
  const long grid = (long)((a->N + 15) >> 4) * a->splitk;
  if (grid > 0x7fffffffL) return -2;
  const dim3 g((unsigned)grid), b(256);
  if (a->M == 1) hipLaunchKernelGGL(vpt_gemv_kernel<1>, g, b, 0, stream, *a);
  else if (a->M == 2) hipLaunchKernelGGL(vpt_gemv_kernel<2>, g, b, 0, stream, *a);
  else if (a->M <= 4) hipLaunchKernelGGL(vpt_gemv_kernel<4>, g, b, 0, stream, *a);
  else hipLaunchKernelGGL(vpt_gemv_kernel<8>, g, b, 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
