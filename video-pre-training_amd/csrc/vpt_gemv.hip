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
// ROWS (16 / 4 / 2) is chosen so that every CU holds several workgroups (vpt_gemv_launch): at N = 2048 sixteen rows per workgroup
// put 128 workgroups on 256 CUs with 2 MB of loads in flight chip-wide -- 1 TB/s, 24 us per trunk layer of the T = 1 step.
#include "vpt_common.h"
#include "vpt_kernels.h"

#define GEMV_LN_MAXK 3072   // hidsize of the largest policy (3x); 8 rows x 3072 x 2 B = 48 KB of LDS
#define GEMV_PRE 8          // 16-byte weight loads in flight per lane

// Fused LayerNorm prologue (LN = true): the acting step's LayerNorms all feed a linear layer, and at M = 1 a LayerNorm launch
// is a 10 us link in a serial chain of 60 kernels for 8 KB of data.  Every workgroup normalises the M rows itself -- wave w takes
// rows w, w + 4 with EXACTLY the arithmetic of vpt_layernorm_kernel (one wave per row, same lane -> element map, same butterfly
// sums, same rounding to the 16-bit operand), so the result is bit-identical to the two-kernel path -- into LDS, from where the
// main loop takes its A operand.  The weight stream does not depend on it: the first weight loads are already in flight.
template <int MR, int ROWS, bool LN>
__global__ __launch_bounds__(256) void vpt_gemv_kernel(VptGemmArgs a) {
  constexpr int KPI = 64 / (4 * ROWS);          // k blocks covered by one wave-wide load instruction
  __shared__ float part_[4][ROWS][MR];
  __shared__ __attribute__((aligned(16))) vpt_op16 arow_[LN ? MR * GEMV_LN_MAXK : 8];
  constexpr int XS_ROWS = MR < 4 ? MR : 4;      // fp32 staging rows of the LayerNorm prologue: one per wave that has a row
  __shared__ __attribute__((aligned(16))) float xs_[LN ? XS_ROWS * GEMV_LN_MAXK : 4];
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
  const int kbf = kb0 + w * KPI + ksub;        // this lane's first k block
  // The weight matrix is read exactly once, by one CU: nontemporal loads (MI355X_MICROARCH.md "nt-weights": issued -> landed -18 %),
  // GEMV_PRE of them (128 bytes per lane, 32 KB per workgroup) in flight before anything else happens -- at M = 1 the layer is HBM
  // latency x bytes in flight, nothing else (round 2 kept 4: 1.4-1.9 TB/s on the 26-34 MB layers of the 2x trunk).
  u32x4 wpre[GEMV_PRE];
#pragma unroll
  for (int j = 0; j < GEMV_PRE; ++j) {
    const int kb = kbf + j * 4 * KPI;
    wpre[j] = (u32x4){0u, 0u, 0u, 0u};
    if (kb < kb1) wpre[j] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)kb * 4096));
  }
  if (LN && MR <= 4) {
    // (the weight loads above do not depend on the normalisation: they are in flight while it runs).  Up to four rows (the acting
    // path): the whole workgroup shares the work -- every thread fetches its slice of the rows, of the gain and of the bias at
    // once (one memory round trip), wave m then runs the two statistics passes of row m on the LDS copy with EXACTLY the lane ->
    // element map, operation order and butterfly sums of vpt_layernorm_kernel, and the element-wise apply is spread over all 256
    // threads again: bit-identical to the two-kernel path.  (One wave doing all of it cost 3.8 us per launch, 11 launches a step.)
    const int n4 = a.K >> 2;
    __shared__ float stat_[4][2];
    f32x4 gv[GEMV_LN_MAXK / 1024], bv[GEMV_LN_MAXK / 1024];
#pragma unroll
    for (int q = 0; q < GEMV_LN_MAXK / 1024; ++q) {
      const int i4 = tid + 256 * q;
      if (i4 < n4) {
        gv[q] = *(const f32x4*)(a.ln_gain + 4 * i4);
        bv[q] = *(const f32x4*)(a.ln_bias + 4 * i4);
#pragma unroll
        for (int m = 0; m < MR; ++m)
          if (m < a.M) {
            f32x4 v = *(const f32x4*)(a.ln_x + (size_t)m * a.K + 4 * i4);
            if (a.ln_relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *(f32x4*)(xs_ + (size_t)m * GEMV_LN_MAXK + 4 * i4) = v;
          }
      }
    }
    __syncthreads();
    if (w < a.M) {
      const float* xs = xs_ + (size_t)w * GEMV_LN_MAXK;
      float s = 0.f;
      for (int i4 = lane; i4 < n4; i4 += 64) {
        const f32x4 v = *(const f32x4*)(xs + 4 * i4);
        s += (v.x + v.y) + (v.z + v.w);
      }
      const float mean = wave_sum(s) / (float)a.K;
      float ss = 0.f;
      for (int i4 = lane; i4 < n4; i4 += 64) {
        const f32x4 v = *(const f32x4*)(xs + 4 * i4);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      const float rstd = rsqrtf(wave_sum(ss) / (float)a.K + VPT_NORM_EPS);
      if (lane == 0) { stat_[w][0] = mean; stat_[w][1] = rstd; }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m >= a.M) break;
      const float mean = stat_[m][0], rstd = stat_[m][1];
#pragma unroll
      for (int q = 0; q < GEMV_LN_MAXK / 1024; ++q) {
        const int i4 = tid + 256 * q;
        if (i4 >= n4) continue;
        const f32x4 v = *(const f32x4*)(xs_ + (size_t)m * GEMV_LN_MAXK + 4 * i4);
        const f32x4 g = gv[q], b = bv[q];
        f32x4 y;
        y.x = fmaf((v.x - mean) * rstd, g.x, b.x);
        y.y = fmaf((v.y - mean) * rstd, g.y, b.y);
        y.z = fmaf((v.z - mean) * rstd, g.z, b.z);
        y.w = fmaf((v.w - mean) * rstd, g.w, b.w);
        if (a.ln_out_f32 && blockIdx.x == 0) *(f32x4*)(a.ln_out_f32 + (size_t)m * a.K + 4 * i4) = y;
        u32x2 p = {pack_op16x2(y.x, y.y), pack_op16x2(y.z, y.w)};
        *(u32x2*)(arow_ + (size_t)m * a.K + 4 * i4) = p;
      }
    }
    __syncthreads();
  } else if (LN) {
    // five to eight rows: wave w takes rows w, w + 4 on its own (per-wave staging row), same arithmetic
    const int n4 = a.K >> 2;
    float* xs = xs_ + (size_t)(w % XS_ROWS) * GEMV_LN_MAXK;
    for (int m = w; m < a.M; m += 4) {
      const float* x = a.ln_x + (size_t)m * a.K;
      for (int i = lane; i < n4; i += 64) {
        f32x4 v = *(const f32x4*)(x + 4 * i);
        if (a.ln_relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(f32x4*)(xs + 4 * i) = v;
      }
      float s = 0.f;
      for (int i = lane; i < n4; i += 64) {
        const f32x4 v = *(const f32x4*)(xs + 4 * i);
        s += (v.x + v.y) + (v.z + v.w);
      }
      const float mean = wave_sum(s) / (float)a.K;
      float ss = 0.f;
      for (int i = lane; i < n4; i += 64) {
        const f32x4 v = *(const f32x4*)(xs + 4 * i);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      const float rstd = rsqrtf(wave_sum(ss) / (float)a.K + VPT_NORM_EPS);
      for (int i = lane; i < n4; i += 64) {
        const f32x4 v = *(const f32x4*)(xs + 4 * i);
        const f32x4 g = *(const f32x4*)(a.ln_gain + 4 * i), b = *(const f32x4*)(a.ln_bias + 4 * i);
        f32x4 y;
        y.x = fmaf((v.x - mean) * rstd, g.x, b.x);
        y.y = fmaf((v.y - mean) * rstd, g.y, b.y);
        y.z = fmaf((v.z - mean) * rstd, g.z, b.z);
        y.w = fmaf((v.w - mean) * rstd, g.w, b.w);
        if (a.ln_out_f32 && blockIdx.x == 0) *(f32x4*)(a.ln_out_f32 + (size_t)m * a.K + 4 * i) = y;
        u32x2 p = {pack_op16x2(y.x, y.y), pack_op16x2(y.z, y.w)};
        *(u32x2*)(arow_ + (size_t)m * a.K + 4 * i) = p;
      }
    }
    __syncthreads();
  }
  auto mac = [&](const u32x4& wraw, int kb) {   // 8 MACs per row: four packed-pair dot products, fp32 accumulate, k ascending
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      u32x4 av;
      if (LN) av = *(const u32x4*)(arow_ + (size_t)min(m, a.M - 1) * a.K + kb * 32 + chunk * 8);
      else av = *(const u32x4*)(ap + (size_t)min(m, a.M - 1) * a.lda + kb * 32);
      acc[m] = dot2_op16(wraw.x, av.x, acc[m]);
      acc[m] = dot2_op16(wraw.y, av.y, acc[m]);
      acc[m] = dot2_op16(wraw.z, av.z, acc[m]);
      acc[m] = dot2_op16(wraw.w, av.w, acc[m]);
    }
  };
#pragma unroll
  for (int j = 0; j < GEMV_PRE; ++j)
    if (kbf + j * 4 * KPI < kb1) mac(wpre[j], kbf + j * 4 * KPI);
  for (int kb = kbf + GEMV_PRE * 4 * KPI; kb < kb1; kb += GEMV_PRE * 4 * KPI) {   // further rounds of GEMV_PRE loads per lane (K >= 4096)
#pragma unroll
    for (int j = 0; j < GEMV_PRE; ++j)
      if (kb + j * 4 * KPI < kb1) wpre[j] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)(kb + j * 4 * KPI) * 4096));
#pragma unroll
    for (int j = 0; j < GEMV_PRE; ++j)
      if (kb + j * 4 * KPI < kb1) mac(wpre[j], kb + j * 4 * KPI);
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

static int g_gemv_rows = 0;
extern "C" void vpt_gemv_set_rows(int rows) { g_gemv_rows = rows; }   // profiling hook (tools/gemv_bench.py): force ROWS, 0 = automatic

extern "C" int vpt_gemv_launch(const VptGemmArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->M > 8 || a->N <= 0 || (a->K & 31) || a->splitk < 1) return -1;
  if (!a->ln_x && (a->lda & 7)) return -1;
  if (a->ln_x && (a->K > GEMV_LN_MAXK || a->splitk != 1 || !a->ln_gain || !a->ln_bias)) return -1;
  if (a->splitk > 1 && (!a->atomic_out || a->relu || a->res || a->out_bf16 || a->mask)) return -1;
  // workgroups per CU decide the bytes in flight: ROWS = 16 only when that still leaves >= 8 workgroups per CU (N >= 32768 rows x splits),
  // 2 when four rows would give fewer than 4 per CU (N <= 4096: the trunk's hid -> hid layers)
  const long units = (long)a->N * a->splitk;
  int rows = 2;                                   // the fewest rows per workgroup that still fit the grid into ONE round of
  while (rows < 16 && units > 1280L * rows) rows *= 2;   // resident workgroups (5 per CU at the LayerNorm variant's 96 VGPRs)
  if (g_gemv_rows == 2 || g_gemv_rows == 4 || g_gemv_rows == 8 || g_gemv_rows == 16) rows = g_gemv_rows;
  const long grid = (long)((a->N + rows - 1) / rows) * a->splitk;
  if (grid > 0x7fffffffL) return -2;
  const dim3 g((unsigned)grid), b(256);
#define GEMV__(MR_, LN_) do { if (rows == 16) hipLaunchKernelGGL((vpt_gemv_kernel<MR_, 16, LN_>), g, b, 0, stream, *a); \
                              else if (rows == 8) hipLaunchKernelGGL((vpt_gemv_kernel<MR_, 8, LN_>), g, b, 0, stream, *a); \
                              else if (rows == 4) hipLaunchKernelGGL((vpt_gemv_kernel<MR_, 4, LN_>), g, b, 0, stream, *a); \
                              else hipLaunchKernelGGL((vpt_gemv_kernel<MR_, 2, LN_>), g, b, 0, stream, *a); } while (0)
#define GEMV_(MR_) do { if (a->ln_x) GEMV__(MR_, true); else GEMV__(MR_, false); } while (0)
  if (a->M == 1) GEMV_(1);
  else if (a->M == 2) GEMV_(2);
  else if (a->M <= 4) GEMV_(4);
  else GEMV_(8);
#undef GEMV_
#undef GEMV__
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
