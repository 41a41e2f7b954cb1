// Fixed-order column sums of a partial-sum slab (gfx950): the second stage of every cross-workgroup reduction of the BC backward.
//
// The reference's backward (torch autograd on one device, behavioural_cloning.py:117-122) returns the same bits for the same batch.  The first
// stage of each reduction here leaves ONE row of partial sums per workgroup in a caller-owned slab -- no atomics, so nothing depends on the order in
// which workgroups run -- and this kernel adds the rows of a column in an order that is a function of (rows, cols) alone:
//   rows are cut into slices of SLAB_SLICE rows; inside a slice four contiguous segments are summed front to back by four waves and combined as
//   (s0 + s1) + (s2 + s3); with more than one slice the slice sums go to `scratch` and a second launch adds them the same way.
// out_a receives columns [0, split), out_b columns [split, cols) (two destination tensors of one slab, e.g. dgain / dbias); accumulate: += (the
// destination carries the sum over earlier launches of the same stream -- the frame chunks of one accumulator -- in launch order).
#include "vpt_common.h"
#include "vpt_kernels.h"

#define SLAB_SLICE 256

__global__ __launch_bounds__(256) void vpt_slab_sum_kernel(const float* __restrict__ slab, int rows, int cols, long ld, float* out_a, int split,
                                                           float* out_b, int accumulate, float* scratch) {
  __shared__ float seg_[4][64];
  const int l = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + l;
  const int r0 = blockIdx.y * SLAB_SLICE, r1 = min(r0 + SLAB_SLICE, rows);
  const int per = (r1 - r0 + 3) >> 2, b0 = r0 + sg * per, b1 = min(b0 + per, r1);
  float s = 0.f;
  if (col < cols) {
    const float* p = slab + col;
#pragma unroll 8
    for (int r = b0; r < b1; ++r) s += p[(size_t)r * ld];
  }
  seg_[sg][l] = s;
  __syncthreads();
  if (sg != 0 || col >= cols) return;
  const float tot = (seg_[0][l] + seg_[1][l]) + (seg_[2][l] + seg_[3][l]);
  if (gridDim.y > 1) { scratch[(size_t)blockIdx.y * cols + col] = tot; return; }
  float* dst = (col < split) ? out_a + col : out_b + (col - split);
  *dst = accumulate ? *dst + tot : tot;
}

extern "C" long vpt_slab_sum_scratch_floats(int rows, int cols) {
  return rows > SLAB_SLICE ? (long)((rows + SLAB_SLICE - 1) / SLAB_SLICE) * cols : 0;
}

extern "C" int vpt_slab_sum_launch(const float* slab, int rows, int cols, long ld, float* out_a, int split, float* out_b, int accumulate,
                                   float* scratch, hipStream_t stream) {
  if (rows <= 0 || cols <= 0 || !slab || !out_a || (split < cols && !out_b)) return -1;
  const int slices = (rows + SLAB_SLICE - 1) / SLAB_SLICE;
  if (slices > SLAB_SLICE || (slices > 1 && !scratch)) return -1;
  const dim3 b(256);
  hipLaunchKernelGGL(vpt_slab_sum_kernel, dim3((cols + 63) / 64, slices), b, 0, stream, slab, rows, cols, ld, out_a, split, out_b, accumulate, scratch);
  if (slices > 1)
    hipLaunchKernelGGL(vpt_slab_sum_kernel, dim3((cols + 63) / 64, 1), b, 0, stream, (const float*)scratch, slices, cols, (long)cols, out_a, split, out_b,
                       accumulate, (float*)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
