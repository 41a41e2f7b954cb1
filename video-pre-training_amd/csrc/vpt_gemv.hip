// Skinny linear layer for the acting path (M = B*T <= 8 rows, agent.py:190-206): out[m][n] = sum_k A[m][k] W[n][k]
// with the same fused epilogue as vpt_gemm_kernel (bias, ReLU, gate mask, fp32 residual, fp32 / bf16 outputs, or
// split-K partial slices).  At M = 1 the layer is a stream of the weight matrix (2 bytes per MAC): HBM-bound, so the job is
// to keep many 16-byte loads in flight on every CU rather than to feed the MFMA -- a 256 x 128 GEMM tile would
// leave 1 row of 256 busy and put N/128 workgroups on a 256-CU chip.
//
// Workgroup = ROWS output columns (rows of W) of one packed 128-row tile; the packed layout [NT][K/32][128][32] makes
// ROWS rows x 32 k one contiguous ROWS * 64-byte run; lane = (k block within the instruction, row, 8-k chunk), so one
// 16-byte load per lane covers 64 / (4 ROWS) consecutive k blocks; the 4 waves interleave over the k blocks; fp32 accumulate;
// reduce over the chunk / k-block lanes by shuffles and over the waves through LDS.  grid = ceil(N / ROWS) x splitk.
// ROWS = 16 when that already gives >= 512 workgroups, else 4: at N = 2048 sixteen rows per workgroup put 128 workgroups on
// 256 CUs with 2 MB of loads in flight chip-wide -- 1 TB/s, 24 us per trunk layer of the T = 1 step; with four rows every byte
// of the matrix is requested up front by 512+ workgroups.
#include "vpt_common.h"
#include "vpt_kernels.h"

template <int MR, int ROWS>
__global__ __launch_bounds__(256) void vpt_gemv_kernel(VptGemmArgs a) {
  constexpr int KPI = 64 / (4 * ROWS);          // k blocks covered by one wave-wide load instruction
  __shared__ float part_[4][ROWS][MR];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int NB = (a.N + ROWS - 1) / ROWS;
  const int nb = blockIdx.x % NB, split = blockIdx.x / NB;
  const int r0 = nb * ROWS, nt = r0 >> 7, rin = r0 & 127;
  const int row = (lane >> 2) % ROWS, chunk = lane & 3, ksub = lane / (4 * ROWS);
  const int kbs = a.K >> 5;
  const int per = (kbs + a.splitk - 1) / a.splitk;
  const int kb0 = split * per, kb1 = min(kb0 + per, kbs);
  const vpt_op16* wp = a.wpk + ((size_t)nt * kbs * 128 + rin + row) * 32 + chunk * 8;
  const vpt_op16* ap = a.A + chunk * 8;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
#pragma unroll 4
  for (int kb = kb0 + w * KPI + ksub; kb < kb1; kb += 4 * KPI) {
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
#pragma unroll
    for (int o = 4 * ROWS; o < 64; o <<= 1) acc[m] += __shfl_xor(acc[m], o, 64);   // the k blocks of one instruction
  }
  if (chunk == 0 && ksub == 0) {
#pragma unroll
    for (int m = 0; m < MR; ++m) part_[w][row][m] = acc[m];
  }
  __syncthreads();
  if (tid < ROWS * MR) {
    const int m = tid / ROWS, r = tid % ROWS;     // consecutive threads -> consecutive output columns
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
  const int rows = ((a->N + 15) >> 4) * a->splitk >= 512 ? 16 : 4;
  const long grid = (long)((a->N + rows - 1) / rows) * a->splitk;
  if (grid > 0x7fffffffL) return -2;
  const dim3 g((unsigned)grid), b(256);
#define GEMV_(MR_) do { if (rows == 16) hipLaunchKernelGGL((vpt_gemv_kernel<MR_, 16>), g, b, 0, stream, *a); \
                        else hipLaunchKernelGGL((vpt_gemv_kernel<MR_, 4>), g, b, 0, stream, *a); } while (0)
  if (a->M == 1) GEMV_(1);
  else if (a->M == 2) GEMV_(2);
  else if (a->M <= 4) GEMV_(4);
  else GEMV_(8);
#undef GEMV_
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
