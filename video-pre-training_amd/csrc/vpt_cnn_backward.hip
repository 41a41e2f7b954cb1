// Backward of the HBM-bound pieces of the IMPALA CNN (gfx950): frame-wide affine norms, max-pool, and the
// per-element preparation of a normed conv layer's backward.  All tensors bf16 channel-blocked
// [frame][C/32][H][W][32]; reductions in fp32 registers -> LDS -> the workgroup's row of a partial slab, summed in a fixed order
// (vpt_reduce.hip; the per-frame scalars: fp64 atomics of fp32 workgroup sums, exact and therefore order-free).
//
//  vpt_affine_bwd_reduce / _apply : backward of y = (x - mu_f) rstd_f g + b with whole-frame statistics
//        (CnnDownStack.n = GroupNorm(1,C), lib/impala_cnn.py:99-100,118-119; ImpalaCNN.dense's LayerNorm,
//        lib/impala_cnn.py:177-184):  dx = rstd (dy g - mean_f(dy g) - xhat mean_f(dy g xhat)).
//  vpt_affine_bwd_elem            : dgain / dbias of the per-element (65536-wide) variant, reduced over frames.
//  vpt_pool_bwd                   : F.max_pool2d(3, 2, 1) backward (lib/impala_cnn.py:117) with torch's tie rule
//        (first maximum in window scan order keeps the gradient).
//  vpt_conv_bwd_prep              : for GN -> conv3x3 -> ReLU (+res) with the GroupNorm folded into the epilogue
//        (vpt_conv3x3.hip):  dz = dY * [v > 0],  dacc = rstd_f dz  (operand of the dgrad / wgrad convolutions),
//        T1_f = sum dz (v - SA[e,o]),  T2_f = sum dz SG[e,o]  (gradients w.r.t. rstd_f, mu_f),
//        dSA[e,o] = sum dz,  dSG[e,o] = sum dz (-rstd_f mu_f)  (gradients of the edge tables -> dW, dgain, dbias).
#include "vpt_common.h"
#include "vpt_kernels.h"

#define EW_ITEMS 4
#define EW_PER_BLOCK (256 * EW_ITEMS)

__device__ __forceinline__ void block_sum2_atomic_f64(float a, float b, double* dst) {
  __shared__ float red_[8];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red_[w] = a; red_[4 + w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(dst, (double)((red_[0] + red_[1]) + (red_[2] + red_[3])));
    atomicAdd(dst + 1, (double)((red_[4] + red_[5]) + (red_[6] + red_[7])));
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// pass 1: AB[f] += (sum dy g, sum dy g xhat); per-channel dgain[c] += sum dy xhat, dbias[c] += sum dy.
// One workgroup = up to RED_PIX pixels of one (frame, channel block); thread = (channel octet, pixel phase), so the
// per-channel sums stay in 16 registers and are reduced by shuffles -> LDS -> 64 global atomics per workgroup.
#define RED_PIX 1024
__device__ __forceinline__ float sum_oct16(float v) {  // over the 16 lanes sharing (lane & 3)
#pragma unroll
  for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Deterministic accumulation of a thread column's edge-class sums into the workgroup's table (all 256 threads call it): every wave stages its own
// contribution in LDS, and after one barrier each table entry adds the four waves' values in a FIXED order.  Until the end of round 5 these were LDS
// float atomics: S then depended on the order in which the four waves arrived, and through T1 = (sum dz v) - <SA, S> -- a difference of nearly equal
// numbers -- so did the frame's coefficients (c0, c1): repeating the SAME gradient computation landed on a handful of discrete alternative outcomes
// (3.6e-5 ... 5e-4 on the stack-0 tensors in ~15 % of the runs, once the lighter reduce-only / pooled kernels let the waves finish almost together:
// tools/diag_shards.py, DESIGN.md "Known issue").
// stage_: [4 waves][9 classes][32 channels], zero on entry and zero again on return.  red: the interior columns' sums after the per-wave butterfly
// (lanes 0-3, one per channel octet); top / all / bot: the thread's own sums, used when its column is an edge column (ex != 1).  jrow / J: the thread's
// pixel row inside its wave's 16 pixels and how many rows a wave holds (J > 1 only for images narrower than 16 pixels: two lanes of a wave then share
// an edge entry and add in row order).
#define EDGE_STAGE_FLOATS (4 * 9 * 32)
__device__ __forceinline__ void add_edge_sums_ordered(float* tab_, float* stage_, int ex, int oct, int lane, int wave, int jrow, int J,
                                                      const float (&top)[8], const float (&all)[8], const float (&bot)[8]) {
  float* mine = stage_ + wave * (9 * 32) + oct * 8;
  // one row class (top / interior rows / bottom) at a time, butterfly and store back to back: eight values in flight, not twenty-four
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float (&src)[8] = (g == 0) ? top : ((g == 1) ? all : bot);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float r = sum_oct16((ex == 1) ? src[k] : 0.f);     // interior columns: most lanes (edge-column lanes contribute zero)
      if (lane < 4) mine[(g * 3 + 1) * 32 + k] = r;
    }
    // J > 1 (images narrower than 16 pixels): two lanes of one wave add into the SAME entry, one after the other.  The accesses are volatile and a wave
    // barrier separates the rows, so the compiler can neither keep the entry in a register across the loop nor merge the iterations (either would
    // lose one lane's addend; LDS operations of one wave execute in program order).
    volatile float* vm = mine;
    for (int j = 0; j < J; ++j) {
      if (ex != 1 && jrow == j) {
#pragma unroll
        for (int k = 0; k < 8; ++k) vm[(g * 3 + ex) * 32 + k] = vm[(g * 3 + ex) * 32 + k] + src[k];
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * 32; i += 256) {
    tab_[i] += (stage_[i] + stage_[288 + i]) + (stage_[576 + i] + stage_[864 + i]);
    stage_[i] = 0.f; stage_[288 + i] = 0.f; stage_[576 + i] = 0.f; stage_[864 + i] = 0.f;
  }
  __syncthreads();
}

template <bool PER_ELEMENT>
__global__ __launch_bounds__(256) void vpt_affine_bwd_reduce_kernel(VptAffineBwdArgs a) {
  __shared__ float part_[4 * 64];
  const int chunks = (a.HW + RED_PIX - 1) / RED_PIX;
  int L = blockIdx.x;
  const int chunk = L % chunks; L /= chunks;
  const int cb = L % a.CB, f = L / a.CB;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
  const int oct = threadIdx.x & 3;
  const size_t fbase = (size_t)f * a.CB * a.HW * 32;
  float g[8], dgc[8], dbc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    g[k] = PER_ELEMENT ? 0.f : a.gain[cb * 32 + oct * 8 + k];
    dgc[k] = dbc[k] = 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
  const int p_end = min(a.HW, (chunk + 1) * RED_PIX);
  // four pixels per trip: their loads (clamped, so unconditional) all go out before the first is used
  for (int p0 = chunk * RED_PIX + (threadIdx.x >> 2); p0 < p_end; p0 += 256) {
    u32x4 xv[4], dv[4];
    f32x4 gv[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pix = min(p0 + 64 * j, p_end - 1);
      const size_t eoff = ((size_t)cb * a.HW + pix) * 32 + oct * 8;
      xv[j] = VPT_LD_STREAM((const u32x4*)(a.x + fbase + eoff));
      dv[j] = VPT_LD_STREAM((const u32x4*)(a.dy + fbase + eoff));
      if (PER_ELEMENT) { gv[j][0] = *(const f32x4*)(a.gain + eoff); gv[j][1] = *(const f32x4*)(a.gain + eoff + 4); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = p0 + 64 * j < p_end;
      float x[8], dy[8];
      unpack8(xv[j], x);
      unpack8(dv[j], dy);
      if (PER_ELEMENT) {
        g[0] = gv[j][0].x; g[1] = gv[j][0].y; g[2] = gv[j][0].z; g[3] = gv[j][0].w;
        g[4] = gv[j][1].x; g[5] = gv[j][1].y; g[6] = gv[j][1].z; g[7] = gv[j][1].w;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = live ? dy[k] : 0.f;   // a clamped duplicate contributes nothing
        const float xh = (x[k] - mean) * rstd, dg = d * g[k];
        s1 += dg;
        s2 = fmaf(dg, xh, s2);
        dgc[k] = fmaf(d, xh, dgc[k]);
        dbc[k] += d;
      }
    }
  }
  block_sum2_atomic_f64(s1, s2, a.ab + 2 * f);
  if (!PER_ELEMENT) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dgc[k] = sum_oct16(dgc[k]);
      dbc[k] = sum_oct16(dbc[k]);
    }
    if (lane < 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        part_[w * 64 + lane * 8 + k] = dgc[k];
        part_[w * 64 + 32 + lane * 8 + k] = dbc[k];
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // this workgroup's entries of slab row (frame, pixel chunk): [dgain C | dbias C]; the launcher's vpt_slab_sum adds the rows in order
      const float v = (part_[threadIdx.x] + part_[64 + threadIdx.x]) + (part_[128 + threadIdx.x] + part_[192 + threadIdx.x]);
      const int C = a.CB * 32;
      a.partials[((size_t)f * chunks + chunk) * (2 * C) + (threadIdx.x < 32 ? 0 : C) + cb * 32 + (threadIdx.x & 31)] = v;
    }
  }
}

// pass 2: dx = rstd (dy g - A/n - xhat B/n) [+ dx_add]
template <bool PER_ELEMENT, bool HAS_ADD>
__global__ __launch_bounds__(256) void vpt_affine_bwd_apply_kernel(VptAffineBwdArgs a) {
  const int per_frame = a.CB * a.HW * 4;
  const int blocks_per_frame = (per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK;
  const int f = blockIdx.x / blocks_per_frame;
  const int base = (blockIdx.x - f * blocks_per_frame) * EW_PER_BLOCK + threadIdx.x;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
  const float A = (float)(a.ab[2 * f] * a.inv_count), B = (float)(a.ab[2 * f + 1] * a.inv_count);
  // all loads of the thread's EW_ITEMS items first (clamped index, so unconditional), then the arithmetic
  u32x4 xv[EW_ITEMS], dv[EW_ITEMS], ev[EW_ITEMS];
  f32x4 g0[EW_ITEMS], g1[EW_ITEMS];
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = min(base + it * 256, per_frame - 1);
    const size_t off = (size_t)f * per_frame * 8 + (size_t)item * 8;
    xv[it] = VPT_LD_STREAM((const u32x4*)(a.x + off));
    dv[it] = VPT_LD_STREAM((const u32x4*)(a.dy + off));
    if (HAS_ADD) ev[it] = VPT_LD_STREAM((const u32x4*)(a.dx_add + off));
    const int gidx = PER_ELEMENT ? item * 8 : (item / (a.HW * 4)) * 32 + (item & 3) * 8;
    g0[it] = *(const f32x4*)(a.gain + gidx); g1[it] = *(const f32x4*)(a.gain + gidx + 4);
  }
#pragma unroll
  for (int it = 0; it < EW_ITEMS; ++it) {
    const int item = base + it * 256;
    if (item >= per_frame) break;
    const size_t off = (size_t)f * per_frame * 8 + (size_t)item * 8;
    float x[8], dy[8], o[8];
    unpack8(xv[it], x);
    unpack8(dv[it], dy);
    const float g[8] = {g0[it].x, g0[it].y, g0[it].z, g0[it].w, g1[it].x, g1[it].y, g1[it].z, g1[it].w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (x[k] - mean) * rstd;
      o[k] = rstd * (dy[k] * g[k] - A - xh * B);
    }
    if (HAS_ADD) {
      float e[8];
      unpack8(ev[it], e);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += e[k];
    }
    VPT_ST_STREAM(pack8(o), (u32x4*)(a.dx + off));
  }
}

// per-element gain: dgain[i] += sum_f dy xhat, dbias[i] += sum_f dy  (thread = 8 elements, grid.y = frame chunks)
__global__ __launch_bounds__(256) void vpt_affine_bwd_elem_kernel(VptAffineBwdArgs a) {
  const int per_frame = a.CB * a.HW * 4;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= per_frame) return;
  const int fper = (a.frames + gridDim.y - 1) / gridDim.y;
  const int f0 = blockIdx.y * fper, f1 = min(f0 + fper, a.frames);
  float dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, db[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int f = f0; f < f1; ++f) {
    float mean, rstd;
    frame_mean_rstd(a.stats_in, f, a.inv_count, mean, rstd);
    const size_t off = (size_t)f * per_frame * 8 + (size_t)item * 8;
    float x[8], dy[8];
    unpack8(*(const u32x4*)(a.x + off), x);
    unpack8(*(const u32x4*)(a.dy + off), dy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dg[k] = fmaf(dy[k], (x[k] - mean) * rstd, dg[k]);
      db[k] += dy[k];
    }
  }
  // slab row = this frame slice: [dgain K | dbias K], K = elements per frame; the launcher's vpt_slab_sum adds the slices in order
  float* row = a.partials + (size_t)blockIdx.y * (2 * (size_t)per_frame * 8);
  *(f32x4*)(row + (size_t)item * 8) = (f32x4){dg[0], dg[1], dg[2], dg[3]};
  *(f32x4*)(row + (size_t)item * 8 + 4) = (f32x4){dg[4], dg[5], dg[6], dg[7]};
  row += (size_t)per_frame * 8;
  *(f32x4*)(row + (size_t)item * 8) = (f32x4){db[0], db[1], db[2], db[3]};
  *(f32x4*)(row + (size_t)item * 8 + 4) = (f32x4){db[4], db[5], db[6], db[7]};
}

static int affine_elem_slices(int frames) { return frames >= 512 ? 32 : (frames >= 16 ? 8 : 1); }

// floats of `partials` a pass needs (pass 2 none): the slab rows followed by vpt_slab_sum's scratch
extern "C" long vpt_affine_bwd_partial_floats(int frames, int CB, int HW, int per_element, int pass) {
  if (pass == 1 && !per_element) {
    const long rows = (long)frames * ((HW + RED_PIX - 1) / RED_PIX);
    return rows * 2 * CB * 32 + vpt_slab_sum_scratch_floats((int)rows, 2 * CB * 32);
  }
  if (pass == 3) return (long)affine_elem_slices(frames) * 2 * CB * 32 * HW;
  return 0;
}

extern "C" int vpt_affine_bwd_launch(const VptAffineBwdArgs* a, int pass, hipStream_t stream) {
  if (a->frames <= 0) return -1;
  const int per_frame = a->CB * a->HW * 4;
  const long grid = (long)a->frames * ((per_frame + EW_PER_BLOCK - 1) / EW_PER_BLOCK);
  if (grid > 0x7fffffffL) return -2;
  if (pass == 1) {
    const int chunks = (a->HW + RED_PIX - 1) / RED_PIX;
    const long g1 = (long)a->frames * a->CB * chunks;
    if (g1 > 0x7fffffffL) return -2;
    if (a->per_element) hipLaunchKernelGGL(vpt_affine_bwd_reduce_kernel<true>, dim3((unsigned)g1), dim3(256), 0, stream, *a);
    else {
      const long rows = (long)a->frames * chunks;
      const int C2 = 2 * a->CB * 32;
      if (!a->partials || !a->dgain || !a->dbias || rows > 65536) return -1;
      hipLaunchKernelGGL(vpt_affine_bwd_reduce_kernel<false>, dim3((unsigned)g1), dim3(256), 0, stream, *a);
      if (hipGetLastError() != hipSuccess) return -3;
      return vpt_slab_sum_launch(a->partials, (int)rows, C2, C2, a->dgain, C2 / 2, a->dbias, 1, a->partials + rows * C2, stream);
    }
  }
  else if (pass == 2) {
    const dim3 g((unsigned)grid), b(256);
    if (a->per_element) {
      if (a->dx_add) hipLaunchKernelGGL((vpt_affine_bwd_apply_kernel<true, true>), g, b, 0, stream, *a);
      else hipLaunchKernelGGL((vpt_affine_bwd_apply_kernel<true, false>), g, b, 0, stream, *a);
    } else {
      if (a->dx_add) hipLaunchKernelGGL((vpt_affine_bwd_apply_kernel<false, true>), g, b, 0, stream, *a);
      else hipLaunchKernelGGL((vpt_affine_bwd_apply_kernel<false, false>), g, b, 0, stream, *a);
    }
  }
  else {
    const int gy = affine_elem_slices(a->frames);
    const long K2 = 2L * per_frame * 8;
    if (!a->partials || K2 > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(vpt_affine_bwd_elem_kernel, dim3((per_frame + 255) / 256, gy), dim3(256), 0, stream, *a);
    if (hipGetLastError() != hipSuccess) return -3;
    return vpt_slab_sum_launch(a->partials, gy, (int)K2, K2, a->dgain, (int)(K2 / 2), a->dbias, 1, nullptr, stream);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void vpt_pool_bwd_kernel(VptPoolBwdArgs a) {
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int per_frame = a.CB * a.H * a.W * 4;
  const long item_g = (long)blockIdx.x * 256 + threadIdx.x;
  if (item_g >= (long)a.frames * per_frame) return;
  const int f = (int)(item_g / per_frame);
  int r = (int)(item_g - (long)f * per_frame);
  const int oct = r & 3; r >>= 2;
  const int x = r % a.W; r /= a.W;
  const int y = r % a.H;
  const int cb = r / a.H;
  const vpt_op16* pre = a.pre + ((size_t)(f * a.CB + cb) * a.H * a.W) * 32 + oct * 8;
  const size_t pplane = ((size_t)(f * a.CB + cb) * PH * PW) * 32 + oct * 8;
  const u16x8 mine = *(const u16x8*)(pre + (size_t)(y * a.W + x) * 32);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // windows (py, px) that contain (y, x):  2p-1 <= coord <= 2p+1
  const int py_lo = y >> 1, py_hi = (y + 1) >> 1, px_lo = x >> 1, px_hi = (x + 1) >> 1;
  for (int py = py_lo; py <= py_hi; ++py) {
    if (py >= PH) continue;
    for (int px = px_lo; px <= px_hi; ++px) {
      if (px >= PW) continue;
      const u16x8 pm = *(const u16x8*)(a.pooled + pplane + (size_t)(py * PW + px) * 32);
      float d[8];
      unpack8(*(const u32x4*)(a.dpooled + pplane + (size_t)(py * PW + px) * 32), d);
      // is (y, x) the FIRST position of the window (row-major scan) whose value equals the maximum?
      bool first_max[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) first_max[k] = (mine[k] == pm[k]);
      for (int wy = 2 * py - 1; wy <= 2 * py + 1; ++wy) {
        if (wy < 0 || wy >= a.H) continue;
        for (int wx = 2 * px - 1; wx <= 2 * px + 1; ++wx) {
          if (wx < 0 || wx >= a.W) continue;
          if (wy > y || (wy == y && wx >= x)) continue;  // only earlier positions
          const u16x8 o = *(const u16x8*)(pre + (size_t)(wy * a.W + wx) * 32);
#pragma unroll
          for (int k = 0; k < 8; ++k) first_max[k] = first_max[k] && (o[k] != pm[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += first_max[k] ? d[k] : 0.f;
    }
  }
  *(u32x4*)(a.dpre + ((size_t)(f * a.CB + cb) * a.H * a.W + (size_t)(y * a.W + x)) * 32 + oct * 8) = pack8(acc);
}

extern "C" int vpt_pool_bwd_launch(const VptPoolBwdArgs* a, hipStream_t stream) {
  if ((a->H & 1) || (a->W & 1) || a->frames <= 0) return -1;
  const long items = (long)a->frames * a->CB * a->H * a->W * 4;
  const long grid = (items + 255) / 256;
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_pool_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// vpt_conv_bwd_prep: one workgroup = one (frame, 32-channel block) plane.  Thread = (channel octet, image column,
// row phase): the column's edge class is a thread constant and the row's is a loop-position test, so the nine
// per-edge-class sums S[e][o] = sum dz live in 24 registers (all rows / first row / last row) and reach LDS once
// per thread.  Everything that is linear in S is finished by vpt_conv_bwd_finish_kernel from the per-frame
// partials:  dSA = sum_f S_f,  dSG = sum_f (-rstd_f mu_f) S_f,  T1_f = sum dz v - <SA, S_f>,  T2_f = <SG, S_f>.
// With (dpooled, argmax) instead of dy the max-pool backward is fused in: dy(y,x) = sum over the <= 4 windows that
// contain (y,x) of dpooled[window] * [argmax[window] == position code of (y,x)].

// Loads are issued one row ahead of their use (the variants are compile-time, so no branch sits between a load and
// the next): per thread 32..112 bytes in flight, which is what keeps a workgroup-per-plane walk at HBM speed.
struct PrepRow {
  u32x4 y, dy, res;
  u32x4 dp[4];
  uint64_t am[4];
};

// PRE (round 5): `dy` is the operand dacc = rstd dz ITSELF, written by the producing dgrad's epilogue (vpt_conv3x3_kernel mode 6: a block's
// conv1 -> conv0, no residual): one tensor read, nothing written; the sums are those of dacc, scaled back by 1 / rstd when they are
// stored, and the data term sum dz v arrives per frame (gate_u).
template <bool HAS_DY, bool HAS_RES, bool PRE = false>
__global__ __launch_bounds__(256) void vpt_conv_bwd_prep_kernel(VptConvBwdPrepArgs a) {
  static_assert(!PRE || (HAS_DY && !HAS_RES), "the pre-gated variant reads dacc only");
  __shared__ float tab_[9 * 32];
  __shared__ float stage_[EDGE_STAGE_FLOATS];
  const int HW = a.H * a.W;
  const int cb = blockIdx.x % a.CB, f = blockIdx.x / a.CB;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
  for (int i = threadIdx.x; i < 9 * 32; i += 256) tab_[i] = 0.f;
  for (int i = threadIdx.x; i < EDGE_STAGE_FLOATS; i += 256) stage_[i] = 0.f;
  __syncthreads();
  const int oct = threadIdx.x & 3, pi = threadIdx.x >> 2;
  const int x = pi & (a.W - 1), ry = pi >> a.wshift, R = 64 >> a.wshift;
  const size_t plane = ((size_t)(f * a.CB + cb) * HW) * 32 + oct * 8;
  const int PH = a.H >> 1, PW = a.W >> 1;
  const size_t pplane = ((size_t)(f * a.CB + cb) * PH * PW) * 32 + oct * 8;
  // the <= 2 x 2 pooling windows that contain (y, x): A = coord >> 1 (always), B = (coord + 1) >> 1 (odd coord, inside)
  const int pxA = x >> 1, pxB = min((x + 1) >> 1, PW - 1);
  const bool vxB = (x & 1) && ((x + 1) >> 1) < PW;
  const unsigned cxA = (unsigned)(x & 1) + 1u;   // window column code of x: x - 2 px + 1 (B: always 0)
  auto load_row = [&](int y, PrepRow& r) {
    const size_t off = plane + (size_t)(y * a.W + x) * 32;
    if (!PRE) r.y = VPT_LD_STREAM((const u32x4*)(a.y + off));
    if (HAS_RES) r.res = VPT_LD_STREAM((const u32x4*)(a.res + off));
    if (HAS_DY) {
      r.dy = VPT_LD_STREAM((const u32x4*)(a.dy + off));
    } else {
      const int pyA = y >> 1, pyB = min((y + 1) >> 1, PH - 1);
      const size_t p0 = pplane + (size_t)(pyA * PW + pxA) * 32, p1 = pplane + (size_t)(pyA * PW + pxB) * 32;
      const size_t p2 = pplane + (size_t)(pyB * PW + pxA) * 32, p3 = pplane + (size_t)(pyB * PW + pxB) * 32;
      r.am[0] = *(const uint64_t*)(a.argmax + p0); r.dp[0] = *(const u32x4*)(a.dpooled + p0);
      r.am[1] = *(const uint64_t*)(a.argmax + p1); r.dp[1] = *(const u32x4*)(a.dpooled + p1);
      r.am[2] = *(const uint64_t*)(a.argmax + p2); r.dp[2] = *(const u32x4*)(a.dpooled + p2);
      r.am[3] = *(const uint64_t*)(a.argmax + p3); r.dp[3] = *(const u32x4*)(a.dpooled + p3);
    }
  };
  float all[8], top[8], bot[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) all[k] = top[k] = bot[k] = 0.f;
  float tv = 0.f;
  auto process = [&](int y, const PrepRow& cur) {
    float dy[8], v[8], o[8];
    if constexpr (PRE) {
      unpack8(cur.dy, dy);
      const bool is_top = (y == 0), is_bot = (y == a.H - 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        all[k] += dy[k];
        top[k] = is_top ? dy[k] : top[k];
        bot[k] = is_bot ? dy[k] : bot[k];
      }
      return;
    }
    unpack8(cur.y, v);
    if (HAS_DY) {
      unpack8(cur.dy, dy);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) dy[k] = 0.f;
      const bool vyB = (y & 1) && ((y + 1) >> 1) < PH;
      const unsigned cyA = ((unsigned)(y & 1) + 1u) * 3u;
      const unsigned code[4] = {cyA + cxA, cyA, cxA, 0u};
      const bool valid[4] = {true, vxB, vyB, vxB && vyB};
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // same order as the window scan (py, px ascending)
        float d[8];
        unpack8(cur.dp[q], d);
        const unsigned want = valid[q] ? code[q] : 0xffu;   // 0xff matches no stored code (0..8, 15)
#pragma unroll
        for (int k = 0; k < 8; ++k) dy[k] += (((unsigned)(cur.am[q] >> (8 * k)) & 0xffu) == want) ? d[k] : 0.f;
      }
    }
    if (HAS_RES) {
      float rr[8];
      unpack8(cur.res, rr);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] -= rr[k];
    }
    const bool is_top = (y == 0), is_bot = (y == a.H - 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dz = (v[k] > 0.f) ? dy[k] : 0.f;
      o[k] = dz * rstd;
      tv = fmaf(dz, v[k], tv);
      all[k] += dz;
      top[k] = is_top ? dz : top[k];
      bot[k] = is_bot ? dz : bot[k];
    }
    VPT_ST_STREAM(pack8(o), (u32x4*)(a.dacc + plane + (size_t)(y * a.W + x) * 32));
  };
  // two row buffers in ping-pong, so a row's loads are issued a full iteration before their first use and no
  // register copy forces an early wait
  PrepRow ra, rb;
  if (ry < a.H) load_row(ry, ra);
  for (int y = ry; y < a.H; y += 2 * R) {
    load_row(min(y + R, a.H - 1), rb);       // (clamped: past the end the row is re-read, not used)
    __builtin_amdgcn_sched_barrier(0);       // keep the loads ABOVE the arithmetic of the other buffer
    process(y, ra);
    __builtin_amdgcn_sched_barrier(0);
    load_row(min(y + 2 * R, a.H - 1), ra);
    __builtin_amdgcn_sched_barrier(0);
    if (y + R < a.H) process(y + R, rb);
  }
  // S[ey][ex][channel]: reduce over the threads of this column class
  const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
  if (PRE) {       // sums of dacc = rstd dz -> sums of dz
    const float inv = 1.0f / rstd;
#pragma unroll
    for (int k = 0; k < 8; ++k) { all[k] *= inv; top[k] *= inv; bot[k] *= inv; }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) all[k] -= top[k] + bot[k];
  const int lane = threadIdx.x & 63;
  add_edge_sums_ordered(tab_, stage_, ex, oct, lane, threadIdx.x >> 6, (pi & 15) >> a.wshift, max(1, 16 >> a.wshift), top, all, bot);
  // sum dz v of this plane -> sbuf column 9*Cout + cb (the finish kernel adds the planes and the <SA, S> correction)
  tv = wave_sum(tv);
  __shared__ float red_[4];
  if (lane == 0) red_[threadIdx.x >> 6] = tv;
  __syncthreads();
  const int Cout = a.CB * 32;
  float* srow = a.sbuf + (size_t)f * (9 * Cout + a.CB);
  if (threadIdx.x == 0) {
    if (PRE) srow[9 * Cout + cb] = (cb == 0) ? (float)(a.gate_u[f] / (double)rstd) : 0.f;     // the frame's sum dz v, once
    else srow[9 * Cout + cb] = (red_[0] + red_[1]) + (red_[2] + red_[3]);
  }
  for (int i = threadIdx.x; i < 9 * 32; i += 256) srow[(i >> 5) * Cout + cb * 32 + (i & 31)] = tab_[i];
}

// ------------------------------------------------------------------------------------------------
// vpt_conv_bwd_prep_pooled (round 5): the same preparation for the layer in front of the max-pool when its forward was the pool-fused
// convolution with arg-max masks (vpt_conv3x3_kernel mode 7) -- no pre-pool tensor, no arg-max bytes.  Inputs at POOLED resolution:
// dpooled, the pooled tensor P (the ReLU gate is [P > 0] and the layer's value at the arg-max is P itself) and the 9-bit masks.
// Thread = (channel octet, pooled column, row phase) OWNS pooled pixel (py, px) and the 2 x 2 block of pre-pool pixels (2 py + a, 2 px + b):
// every pooled value is loaded once (the gather form above reads each four times), turned into (gated gradient, arg-max position) and
// shared with the three neighbours that own parts of its window through LDS.  Pre-pool pixel (2py+a, 2px+b) collects
//     (0,0): window (py,px) position 4          (0,1): (py,px) 5 + (py,px+1) 3
//     (1,0): (py,px) 7 + (py+1,px) 1            (1,1): (py,px) 8 + (py,px+1) 6 + (py+1,px) 2 + (py+1,px+1) 0
// -- positions 0..3 and 6 of a window lie in the blocks of the threads above / to the left.  Rows are processed in passes of 64 / PW pooled
// rows; a pass's entries go to one of two LDS buffers, the row below the pass's last row comes from the next pass's buffer.
struct PoolEntry { u32x4 g; uint32_t codes; };   // gated gradient (8 x 16 bit), arg-max position per channel (8 x 4 bit)

template <bool NFOLD>
__global__ __launch_bounds__(256, 3) void vpt_conv_bwd_prep_pooled_kernel(VptConvBwdPrepArgs a) {
  __shared__ float tab_[9 * 32];
  __shared__ float stage_[EDGE_STAGE_FLOATS];
  __shared__ __attribute__((aligned(16))) u32x4 gbuf_[2][64][4];
  __shared__ uint32_t cbuf_[2][64][4];
  __shared__ float red_[4];
  const int PH = a.H >> 1, PW = a.W >> 1, pwshift = a.wshift - 1;
  const int cb = blockIdx.x % a.CB, f = blockIdx.x / a.CB;
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
  for (int i = threadIdx.x; i < 9 * 32; i += 256) tab_[i] = 0.f;
  for (int i = threadIdx.x; i < EDGE_STAGE_FLOATS; i += 256) stage_[i] = 0.f;
  const int oct = threadIdx.x & 3, slot = threadIdx.x >> 2;
  const int px = slot & (PW - 1), ph = slot >> pwshift, R = 64 >> pwshift;     // R pooled rows per pass
  const int NP = PH / R;
  const size_t pplane = ((size_t)(f * a.CB + cb) * PH * PW) * 32 + oct * 8;
  const size_t plane = ((size_t)(f * a.CB + cb) * a.H * a.W) * 32 + oct * 8;
  float allE[8], allO[8], topE[8], topO[8], botE[8], botO[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) allE[k] = allO[k] = topE[k] = topO[k] = botE[k] = botO[k] = 0.f;
  float tv = 0.f;
  // NFOLD: the backward of GroupNorm `n` (x = (P - mu_P) r_P gain + bias) applied to the incoming gradient G, exactly vpt_affine_bwd_apply_kernel's
  // arithmetic and rounding point (one 16-bit rounding of d(pooled))
  // (the eight gains of the thread's channel octet are read from LDS where they are used: held in registers across the pass loop they were the
  // eight registers that pushed this variant one over the three-waves-per-SIMD budget -- one spilled register, 8 bytes of scratch)
  __shared__ __attribute__((aligned(16))) float ng_[32];
  float mp = 0.f, rpool = 1.f, nA = 0.f, nB = 0.f;
  if (NFOLD) {
    frame_mean_rstd(a.pool_stats, f, a.inv_count_pool, mp, rpool);
    nA = (float)(a.pool_ab[2 * f] * a.inv_count_pool);
    nB = (float)(a.pool_ab[2 * f + 1] * a.inv_count_pool);
    if (threadIdx.x < 32) ng_[threadIdx.x] = a.n_gain[cb * 32 + threadIdx.x];
    __syncthreads();
  }
  u32x4 rd, rp, rm;                         // this thread's pooled pixel of the pass being loaded: gradient, pooled value, mask
  auto load_pass = [&](int p) {
    const size_t off = pplane + (size_t)((p * R + ph) * PW + px) * 32;
    rd = VPT_LD_STREAM((const u32x4*)(a.dpooled + off));
    rp = VPT_LD_STREAM((const u32x4*)(a.pooled + off));
    rm = VPT_LD_STREAM((const u32x4*)(a.pool_mask + off));
  };
  auto make_entry = [&](int p) -> PoolEntry {     // the thread's own entry stays in registers; the neighbours read the LDS copy
    PoolEntry e;
    float df[8], pf[8];
    unpack8(rd, df);
    unpack8(rp, pf);
    if (NFOLD) {
      const f32x4 g0 = *(const f32x4*)(ng_ + oct * 8), g1 = *(const f32x4*)(ng_ + oct * 8 + 4);
      const float ng[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (pf[k] - mp) * rpool;
        df[k] = rpool * (df[k] * ng[k] - nA - xh * nB);
      }
      rd = pack8(df);          // d(pooled), rounded to 16 bits where the separate pass stored it
      unpack8(rd, df);
    }
    uint32_t codes = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t pw_ = rp[j], mw = rm[j];
      const bool o0 = (short)(pw_ & 0xffffu) > 0, o1 = (short)(pw_ >> 16) > 0;          // ReLU gate: the pooled value is the layer's output at the arg-max
      e.g[j] = rd[j] & ((o0 ? 0xffffu : 0u) | (o1 ? 0xffff0000u : 0u));
      tv = fmaf(o0 ? df[2 * j] : 0.f, pf[2 * j], tv);
      tv = fmaf(o1 ? df[2 * j + 1] : 0.f, pf[2 * j + 1], tv);
      // first position (scan order) that holds the maximum = highest zero bit of the 9-bit mask
      const uint32_t i0 = ~mw & 0x1ffu, i1 = ~(mw >> 16) & 0x1ffu;
      const uint32_t k0 = (uint32_t)__builtin_clz(i0 | 1u) - 23u, k1 = (uint32_t)__builtin_clz(i1 | 1u) - 23u;   // (| 1: a mask without a zero bit cannot occur; stay defined)
      codes |= (k0 | (k1 << 4)) << (8 * j);
    }
    e.codes = codes;
    gbuf_[p & 1][slot][oct] = e.g;
    cbuf_[p & 1][slot][oct] = codes;
    return e;
  };
  // contribution of entry (g, codes) to the pre-pool pixel where it sits at window position `pos`
  auto add_if = [&](float (&dz)[8], const u32x4& g, uint32_t codes, uint32_t pos) {
    float gf[8];
    unpack8(g, gf);
#pragma unroll
    for (int k = 0; k < 8; ++k) dz[k] += (((codes >> (4 * k)) & 15u) == pos) ? gf[k] : 0.f;
  };
  auto emit_pass = [&](int p, const PoolEntry& e00) {
    const int py = p * R + ph;
    const bool has_r = px + 1 < PW, has_b = py + 1 < PH;
    // neighbours: right (same row), below / below-right (next row: the next slot row of this pass, or row 0 of the next pass)
    const int bb = (ph + 1 < R) ? (p & 1) : ((p + 1) & 1);
    const int bslot = (ph + 1 < R) ? slot + PW : px;
    u32x4 g01 = {0u, 0u, 0u, 0u}, g10 = g01, g11 = g01;
    uint32_t c01 = 0xffffffffu, c10 = 0xffffffffu, c11 = 0xffffffffu;                 // position 15 matches nothing
    if (has_r) { g01 = gbuf_[p & 1][slot + 1][oct]; c01 = cbuf_[p & 1][slot + 1][oct]; }
    if (has_b) { g10 = gbuf_[bb][bslot][oct]; c10 = cbuf_[bb][bslot][oct]; }
    if (has_r && has_b) { g11 = gbuf_[bb][bslot + 1][oct]; c11 = cbuf_[bb][bslot + 1][oct]; }
    float d00[8], d01[8], d10[8], d11[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) d00[k] = d01[k] = d10[k] = d11[k] = 0.f;
    add_if(d00, e00.g, e00.codes, 4u);
    add_if(d01, e00.g, e00.codes, 5u); add_if(d01, g01, c01, 3u);
    add_if(d10, e00.g, e00.codes, 7u); add_if(d10, g10, c10, 1u);
    add_if(d11, e00.g, e00.codes, 8u); add_if(d11, g01, c01, 6u); add_if(d11, g10, c10, 2u); add_if(d11, g11, c11, 0u);
    const bool is_top = (py == 0), is_bot = (py == PH - 1);
    float o00[8], o01[8], o10[8], o11[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      allE[k] += d00[k] + d10[k];
      allO[k] += d01[k] + d11[k];
      topE[k] = is_top ? d00[k] : topE[k];
      topO[k] = is_top ? d01[k] : topO[k];
      botE[k] = is_bot ? d10[k] : botE[k];
      botO[k] = is_bot ? d11[k] : botO[k];
      o00[k] = d00[k] * rstd; o01[k] = d01[k] * rstd; o10[k] = d10[k] * rstd; o11[k] = d11[k] * rstd;
    }
    // Whole 128-byte lines per store instruction: the four pre-pool pixels X .. X + 3 of a pair of neighbouring pooled columns are 256 contiguous
    // bytes; the even column's lanes own pixels X, X + 1, the odd column's X + 2, X + 3.  One exchange per row between the pair's lanes (lane ^ 4)
    // -- even sends its second pixel and receives the partner's first -- and the first instruction writes (X, X + 1) from all eight lanes, the second
    // (X + 2, X + 3).  (Each thread storing its own two pixels wrote 64-byte halves of every line with both instructions.)
    const bool odd = px & 1;
    vpt_op16* row0 = a.dacc + plane + (size_t)((2 * py) * a.W + 2 * (px & ~1)) * 32;     // pixel X of this row
    auto store_row = [&](const float (&oa)[8], const float (&ob)[8], vpt_op16* base) {
      const u32x4 A = pack8(oa), B = pack8(ob);
      const u32x4 send = odd ? A : B;
      u32x4 recv;
#pragma unroll
      for (int j = 0; j < 4; ++j) recv[j] = (uint32_t)__shfl_xor((int)send[j], 4, 64);
      const u32x4 first = odd ? recv : A;        // line (X, X + 1): even lanes pixel X (own A), odd lanes pixel X + 1 (the even partner's B)
      const u32x4 second = odd ? B : recv;       // line (X + 2, X + 3): even lanes pixel X + 2 (the odd partner's A), odd lanes pixel X + 3 (own B)
      VPT_ST_STREAM(first, (u32x4*)(base + (odd ? 32 : 0)));
      VPT_ST_STREAM(second, (u32x4*)(base + 64 + (odd ? 32 : 0)));
    };
    store_row(o00, o01, row0);
    store_row(o10, o11, row0 + (size_t)a.W * 32);
  };
  load_pass(0);
  PoolEntry cur = make_entry(0), nxt = cur;
  if (NP > 1) load_pass(1);
  __syncthreads();
  for (int p = 0; p < NP; ++p) {
    if (p + 1 < NP) nxt = make_entry(p + 1); // (waits for the loads issued one iteration ago)
    if (p + 2 < NP) load_pass(p + 2);
    __syncthreads();                         // entries of pass p + 1 visible
    emit_pass(p, cur);
    cur = nxt;
    __syncthreads();                         // buffer p & 1 is free for pass p + 2
  }
  // ---- S[ey][ex][channel] of the thread's even / odd pre-pool column, as in the kernel above ----
  const int lane = threadIdx.x & 63;
  auto reduce_col = [&](int ex, float (&all)[8], float (&top)[8], float (&bot)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) all[k] -= top[k] + bot[k];
    add_edge_sums_ordered(tab_, stage_, ex, oct, lane, threadIdx.x >> 6, (slot & 15) >> pwshift, max(1, 16 >> pwshift), top, all, bot);
  };
  reduce_col(px == 0 ? 0 : 1, allE, topE, botE);
  reduce_col(px == PW - 1 ? 2 : 1, allO, topO, botO);
  tv = wave_sum(tv);
  if (lane == 0) red_[threadIdx.x >> 6] = tv;
  __syncthreads();
  const int Cout = a.CB * 32;
  float* srow = a.sbuf + (size_t)f * (9 * Cout + a.CB);
  if (threadIdx.x == 0) srow[9 * Cout + cb] = (red_[0] + red_[1]) + (red_[2] + red_[3]);
  for (int i = threadIdx.x; i < 9 * 32; i += 256) srow[(i >> 5) * Cout + cb * 32 + (i & 31)] = tab_[i];
}

#define FIN_FB 32  // frames per column-sum workgroup
// finish, one launch with two kinds of workgroups (all independent, so the launch is one round of loads deep):
//   blocks [0, ceil(F/4)):  one WAVE per frame: T1, T2 and the (c0, c1) coefficients from that frame's row of sbuf;
//   the rest:               one thread per column (e, c) of the [9][Cout] table x 32 frames: dSA, dSG partial sums -> the block's slab row
//                           (vpt_conv_bwd_sum_kernel adds the rows in order).
__global__ __launch_bounds__(256) void vpt_conv_bwd_finish_kernel(VptConvBwdPrepArgs a) {
  __shared__ float nrm_[FIN_FB];
  const int Cout = a.CB * 32, ncol = 9 * Cout, ld = ncol + a.CB;
  const int nA = (a.frames + 3) >> 2;
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x < nA) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= a.frames) return;
    float mean, rstd;
    frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
    const float* srow = a.sbuf + (size_t)f * ld;
    float t1c = 0.f, t2c = 0.f;
#pragma unroll 4
    for (int i = lane; i < ncol; i += 64) {
      const int e = i / Cout, c = i - e * Cout;
      const float S = srow[i];
      t1c = fmaf(a.edge_sa[e * a.CoutPad + c], S, t1c);
      t2c = fmaf(a.edge_sg[e * a.CoutPad + c], S, t2c);
    }
    t1c = wave_sum(t1c);
    t2c = wave_sum(t2c);
    if (lane == 0) {
      double tv = 0.0;
      for (int cb = 0; cb < a.CB; ++cb) tv += (double)srow[ncol + cb];
      const double T1 = tv - (double)t1c, T2 = (double)t2c;
      if (a.t12) { a.t12[2 * f] = T1; a.t12[2 * f + 1] = T2; }
      // statistics terms of the input gradient: dx += c0 + c1 x,  c1 = -rstd^2 T1 / n,  c0 = -rstd T2 / n - c1 mu
      const double c1 = -(double)rstd * rstd * T1 * a.inv_count_in;
      a.coef[2 * f] = (float)(-(double)rstd * T2 * a.inv_count_in - c1 * mean);
      a.coef[2 * f + 1] = (float)c1;
    }
    return;
  }
  const int b = blockIdx.x - nA, ncb = (ncol + 255) >> 8;
  const int colb = b % ncb, f0 = (b / ncb) * FIN_FB, nf = min(FIN_FB, a.frames - f0);
  if ((int)threadIdx.x < nf) {
    float mean, rstd;
    frame_mean_rstd(a.stats_in, f0 + threadIdx.x, a.inv_count_in, mean, rstd);
    nrm_[threadIdx.x] = -rstd * mean;
  }
  __syncthreads();
  const int i = colb * 256 + threadIdx.x;
  if (i >= ncol) return;
  const float* s = a.sbuf + (size_t)f0 * ld + i;
  float asa = 0.f, asg = 0.f;
#pragma unroll 8
  for (int k = 0; k < nf; ++k) {
    const float S = s[(size_t)k * ld];
    asa += S;
    asg = fmaf(nrm_[k], S, asg);
  }
  // this 32-frame block's partial sums; vpt_conv_bwd_sum_kernel adds the blocks in block order (until round 5: one fp32 atomic per block)
  float* dsum = a.sbuf + (size_t)a.frames * ld + (size_t)(b / ncb) * 2 * ncol;
  dsum[i] = asa;
  dsum[ncol + i] = asg;
}

// dSA[e][c] += sum over the 32-frame blocks of their partial sums, in block order; likewise dSG
__global__ __launch_bounds__(256) void vpt_conv_bwd_sum_kernel(VptConvBwdPrepArgs a) {
  const int Cout = a.CB * 32, ncol = 9 * Cout, ld = ncol + a.CB;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * ncol) return;
  const int nfb = (a.frames + FIN_FB - 1) / FIN_FB;
  const float* p = a.sbuf + (size_t)a.frames * ld + i;
  float s = 0.f;
  for (int fb = 0; fb < nfb; ++fb) s += p[(size_t)fb * 2 * ncol];
  const int j = i < ncol ? i : i - ncol;
  const int e = j / Cout, c = j - e * Cout;
  float* dst = (i < ncol ? a.d_sa : a.d_sg) + e * a.CoutPad + c;
  *dst += s;
}

extern "C" long vpt_conv_bwd_prep_scratch_floats(int frames, int Cout) {
  return (long)frames * (9L * Cout + Cout / 32) + (long)((frames + FIN_FB - 1) / FIN_FB) * 2 * 9 * Cout;
}

extern "C" int vpt_conv_bwd_prep_launch(const VptConvBwdPrepArgs* a0, hipStream_t stream) {
  VptConvBwdPrepArgs a = *a0;
  if (a.frames <= 0 || !a.sbuf || !a.coef) return -1;
  if (a.gate_u && (!a.dy || a.res)) return -1;
  if (a.pooled) {      // pool-fused forward with arg-max masks: the pooled-resolution kernel
    if (a.dy || a.res || !a.dpooled || !a.pool_mask || !a.dacc || a.W < 16 || a.W > 64 || (a.W & (a.W - 1)) || (a.H & 1)) return -1;
    a.wshift = 31 - __builtin_clz((unsigned)a.W);
    const int PW = a.W >> 1, PH = a.H >> 1, R = 64 / PW;
    if (PH % R) return -1;
    const long gridp = (long)a.frames * a.CB;
    if (gridp > 0x7fffffffL) return -2;
    if (a.n_gain) {
      if (!a.pool_stats || !a.pool_ab) return -1;
      hipLaunchKernelGGL(vpt_conv_bwd_prep_pooled_kernel<true>, dim3((unsigned)gridp), dim3(256), 0, stream, a);
    } else hipLaunchKernelGGL(vpt_conv_bwd_prep_pooled_kernel<false>, dim3((unsigned)gridp), dim3(256), 0, stream, a);
    const int fin_blocks_p = (a.frames + 3) / 4 + ((9 * a.CB * 32 + 255) / 256) * ((a.frames + FIN_FB - 1) / FIN_FB);
    hipLaunchKernelGGL(vpt_conv_bwd_finish_kernel, dim3((unsigned)fin_blocks_p), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(vpt_conv_bwd_sum_kernel, dim3((2 * 9 * a.CB * 32 + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  if (a.W < 8 || a.W > 64 || (a.W & (a.W - 1))) return -1;  // column-per-thread mapping: W in {8,16,32,64}
  if (!a.dy && (!a.dpooled || !a.argmax || (a.H & 1) || (a.W & 1))) return -1;
  a.wshift = 31 - __builtin_clz((unsigned)a.W);
  const long grid = (long)a.frames * a.CB;
  if (grid > 0x7fffffffL) return -2;
  if (a.gate_u) hipLaunchKernelGGL((vpt_conv_bwd_prep_kernel<true, false, true>), dim3((unsigned)grid), dim3(256), 0, stream, a);
  else if (a.dy) {
    if (a.res) hipLaunchKernelGGL((vpt_conv_bwd_prep_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((vpt_conv_bwd_prep_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, stream, a);
  } else {
    if (a.res) hipLaunchKernelGGL((vpt_conv_bwd_prep_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((vpt_conv_bwd_prep_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, stream, a);
  }
  const int fin_blocks = (a.frames + 3) / 4 + ((9 * a.CB * 32 + 255) / 256) * ((a.frames + FIN_FB - 1) / FIN_FB);
  hipLaunchKernelGGL(vpt_conv_bwd_finish_kernel, dim3((unsigned)fin_blocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(vpt_conv_bwd_sum_kernel, dim3((2 * 9 * a.CB * 32 + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
