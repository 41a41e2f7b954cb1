// Fused Adam step for the behavioural-cloning fine-tune (gfx950).
//
// Replaces th.optim.Adam(params, lr, weight_decay).step() as configured at behavioural_cloning.py:63-67,122:
// L2-style weight decay (grad += wd * p, NOT AdamW), bias-corrected moments, eps added to sqrt(v_hat):
//     g' = g * grad_scale + wd * p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2
//     p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (torch's single-tensor formulation, so results match torch.optim.Adam to fp32 rounding).  One launch
// updates a whole flat fp32 bucket (the same buckets the gradient all-reduce uses): 4 streams in, 3 out,
// 28 B/parameter -> HBM-bound; float4 accesses, grid-stride.  grad_scale folds the 1/world_size of the
// data-parallel mean (and any loss scaling) into the update.
#include "vpt_common.h"
#include "vpt_kernels.h"

__global__ __launch_bounds__(256) void vpt_adam_kernel(VptAdamArgs a) {
  const size_t n4 = a.n >> 2;
  const float c1 = 1.0f - a.beta1, c2 = 1.0f - a.beta2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 p = *(const f32x4*)(a.p + 4 * i), g = *(const f32x4*)(a.g + 4 * i);
    f32x4 m = *(const f32x4*)(a.m + 4 * i), v = *(const f32x4*)(a.v + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = fmaf(a.weight_decay, p[k], g[k] * a.grad_scale);
      m[k] = fmaf(a.beta1, m[k], c1 * gk);
      v[k] = fmaf(a.beta2, v[k], c2 * gk * gk);
      const float denom = sqrtf(v[k]) * a.inv_sqrt_bc2 + a.eps;
      p[k] -= a.step_size * (m[k] / denom);
    }
    *(f32x4*)(a.p + 4 * i) = p;
    *(f32x4*)(a.m + 4 * i) = m;
    *(f32x4*)(a.v + 4 * i) = v;
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    const float gk = fmaf(a.weight_decay, a.p[i], a.g[i] * a.grad_scale);
    const float m = fmaf(a.beta1, a.m[i], c1 * gk), v = fmaf(a.beta2, a.v[i], c2 * gk * gk);
    a.m[i] = m; a.v[i] = v;
    a.p[i] -= a.step_size * (m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
  }
}

extern "C" int vpt_adam_launch(const VptAdamArgs* a, hipStream_t stream) {
  if (a->n == 0) return 0;
  size_t blocks = ((a->n >> 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(vpt_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- multi-tensor form: ONE launch updates every parameter tensor of the model (129 on the 2x policy; the per-tensor form cost
// 129 launches of ~15 us for 2 ms of HBM traffic).  `table` is a device array of descriptors sorted by first_block; block b
// finds its tensor by binary search and handles 1024 elements of it (256 threads x float4).
__global__ __launch_bounds__(256) void vpt_adam_multi_kernel(const VptAdamTensor* __restrict__ table, int ntensors, VptAdamArgs h) {
  int lo = 0, hi = ntensors - 1;
  const long b = blockIdx.x;
  while (lo < hi) {   // last descriptor with first_block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  if (h.skip_flag && *h.skip_flag) return;   // uniform: the whole launch is a no-op when the scaled gradients overflowed
  const VptAdamTensor t = table[lo];
  const size_t i0 = ((size_t)(b - t.first_block) * 256 + threadIdx.x) * 4;
  const float c1 = 1.0f - h.beta1, c2 = 1.0f - h.beta2;
  if (i0 + 4 <= t.n) {   // 16-byte accesses like vpt_adam_kernel (views may start at any element: unaligned dwordx4 is legal on gfx950)
    f32x4 p = *(const f32x4*)(t.p + i0), g = *(const f32x4*)(t.g + i0);
    f32x4 m = *(const f32x4*)(t.m + i0), v = *(const f32x4*)(t.v + i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = fmaf(h.weight_decay, p[k], g[k] * h.grad_scale);
      m[k] = fmaf(h.beta1, m[k], c1 * gk);
      v[k] = fmaf(h.beta2, v[k], c2 * gk * gk);
      const float denom = sqrtf(v[k]) * h.inv_sqrt_bc2 + h.eps;
      p[k] -= h.step_size * (m[k] / denom);
    }
    *(f32x4*)(t.p + i0) = p;
    *(f32x4*)(t.m + i0) = m;
    *(f32x4*)(t.v + i0) = v;
    return;
  }
  for (size_t i = i0; i < t.n; ++i) {   // the tensor's last (partial) group of four
    const float pk = t.p[i];
    const float gk = fmaf(h.weight_decay, pk, t.g[i] * h.grad_scale);
    const float m = fmaf(h.beta1, t.m[i], c1 * gk), v = fmaf(h.beta2, t.v[i], c2 * gk * gk);
    t.m[i] = m; t.v[i] = v;
    t.p[i] = pk - h.step_size * (m / (sqrtf(v) * h.inv_sqrt_bc2 + h.eps));
  }
}

extern "C" int vpt_adam_multi_launch(const VptAdamTensor* table_dev, int ntensors, long total_blocks, const VptAdamArgs* h, hipStream_t stream) {
  if (ntensors <= 0 || total_blocks <= 0) return 0;
  if (total_blocks > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, table_dev, ntensors, *h);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}


// ---- overflow check of a loss-scaled step (precision = fp16: the 16-bit gradient buffers carry loss_scale x gradient; a value
// beyond 65504 anywhere in the chain arrives here as inf / nan).  Same table and block mapping as vpt_adam_multi_kernel; the flag
// is only ever set (caller zeroes it), so the unordered writes are benign.  th.cuda.amp.GradScaler's found_inf, in one launch.
__global__ __launch_bounds__(256) void vpt_grads_nonfinite_kernel(const VptAdamTensor* __restrict__ table, int ntensors, int* flag) {
  int lo = 0, hi = ntensors - 1;
  const long b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const VptAdamTensor t = table[lo];
  const size_t i0 = ((size_t)(b - t.first_block) * 256 + threadIdx.x) * 4;
  bool bad = false;
  if (i0 + 4 <= t.n) {
    const f32x4 g = *(const f32x4*)(t.g + i0);
    // x - x is 0 for finite x and nan for inf / nan
    const float z = (g.x - g.x) + (g.y - g.y) + (g.z - g.z) + (g.w - g.w);
    bad = !(z == 0.f);
  } else {
    for (size_t i = i0; i < t.n; ++i) { const float g = t.g[i]; bad |= !((g - g) == 0.f); }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1;
}

extern "C" int vpt_grads_nonfinite_launch(const VptAdamTensor* table_dev, int ntensors, long total_blocks, int* flag, hipStream_t stream) {
  if (ntensors <= 0 || total_blocks <= 0) return 0;
  if (total_blocks > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_grads_nonfinite_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, table_dev, ntensors, flag);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
