// Shared device helpers for the VPT hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- the 16-bit MFMA operand / activation storage type ------------------------------------------------
// Default: bfloat16 (north star: "MFMA bf16 tiles").  Built a second time with -DVPT_OPERAND_F16 (libvpt_hip_f16.so,
// PolicyEngine(precision="fp16")) every 16-bit operand is IEEE half instead: same MFMA rate
// (v_mfma_f32_32x32x16_f16), same bytes, 8x finer rounding -- the parity mode of DESIGN.md "Precision".
#ifdef VPT_OPERAND_F16
typedef _Float16 op16_t;
typedef _Float16 op16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 op16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 op16x8 __attribute__((ext_vector_type(8)));
#define VPT_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
typedef __fp16 tr16x4 __attribute__((ext_vector_type(4)));   // element type the f16 transpose-read builtin is declared with
#define VPT_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4f16
#define VPT_OPERAND_NAME "fp16"
#else
typedef __bf16 op16_t;
typedef __bf16 op16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 op16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 op16x8 __attribute__((ext_vector_type(8)));
#define VPT_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
typedef __bf16 tr16x4 __attribute__((ext_vector_type(4)));
#define VPT_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4bf16
#define VPT_OPERAND_NAME "bf16"
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define VPT_NORM_EPS 1e-5f

// ---- 16-bit operand <-> f32 --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_op16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  op16x2 b = __builtin_convertvector(v, op16x2);  // round to nearest even
  return __builtin_bit_cast(uint32_t, b);
}
#ifdef VPT_OPERAND_F16
__device__ __forceinline__ float op16_lo_to_f32(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed & 0xffffu)); }
__device__ __forceinline__ float op16_hi_to_f32(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed >> 16)); }
// two values that are exact in the operand format (bytes 0..255, 0.0, 1.0): any rounding mode will do
__device__ __forceinline__ uint32_t pack_op16x2_exact(float lo, float hi) { return pack_op16x2(lo, hi); }
#else
__device__ __forceinline__ float op16_lo_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float op16_hi_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xffff0000u); }
// exact values: the bf16 is the upper half of the fp32 pattern (one v_perm_b32 per pair)
__device__ __forceinline__ uint32_t pack_op16x2_exact(float lo, float hi) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x07060302u);
}
#endif

// packed pair dot product with fp32 accumulate: c + a.lo * b.lo + a.hi * b.hi (v_dot2c_f32_bf16 / v_dot2c_f32_f16) -- the
// weight-streaming linears of the acting path: one instruction per two MACs, no 16-bit -> fp32 conversions
// clamp(a.x * b + c, 0, 1) per element: v_pk_fma_f32 with the clamp result modifier and a.x broadcast to both halves (op_sel_hi 0).  The compiler
// does not form it (its clamp fold goes through v_max_f32 x, x clamp, which has no packed fp32 form), hence the inline assembly; the operands are
// VALU / long-retired MFMA results at every use (the hazard recogniser does not look inside inline assembly).
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float dot2_op16(uint32_t a, uint32_t b, float c) {
#ifdef VPT_OPERAND_F16
  typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_, a), __builtin_bit_cast(h2_, b), c, false);
#else
  typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_, a), __builtin_bit_cast(b2_, b), c, false);
#endif
}

// ReLU of a packed pair on its bit patterns: one v_pk_max_i16 (positive 16-bit floats order like signed integers, negative ones and -0 are < 0)
__device__ __forceinline__ uint32_t relu_op16x2(uint32_t p) {
  typedef short s16x2_ __attribute__((ext_vector_type(2)));
  const s16x2_ z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_, p), z));
}
#ifdef VPT_OPERAND_F16
#define OP16_ONE2 0x3c003c00u
#else
#define OP16_ONE2 0x3f803f80u
#endif

// LDS transpose read (ds_read_b64_tr_b16): 4 consecutive 16-bit elements of this lane's column
__device__ __forceinline__ op16x4 lds_tr16_read(const unsigned char* p) {
  return __builtin_bit_cast(op16x4, VPT_DS_READ_TR16_B64((__attribute__((address_space(3))) tr16x4*)(p)));
}

// 8 operands (one 16-byte chunk) -> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& c, float* f) {
  f[0] = op16_lo_to_f32(c.x); f[1] = op16_hi_to_f32(c.x);
  f[2] = op16_lo_to_f32(c.y); f[3] = op16_hi_to_f32(c.y);
  f[4] = op16_lo_to_f32(c.z); f[5] = op16_hi_to_f32(c.z);
  f[6] = op16_lo_to_f32(c.w); f[7] = op16_hi_to_f32(c.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 c;
  c.x = pack_op16x2(f[0], f[1]); c.y = pack_op16x2(f[2], f[3]);
  c.z = pack_op16x2(f[4], f[5]); c.w = pack_op16x2(f[6], f[7]);
  return c;
}

// ---- wave / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// per-frame statistics -> (mean, rstd).  stats[2f] = sum, stats[2f+1] = sum of squares (double)
__device__ __forceinline__ void frame_mean_rstd(const double* __restrict__ stats, int f, double inv_count,
                                                float& mean, float& rstd) {
  const double s = stats[2 * f], ss = stats[2 * f + 1];
  const double m = s * inv_count;
  double var = ss * inv_count - m * m;  // fp64: no cancellation problem even when |mean| >> std
  if (var < 0.0) var = 0.0;
  mean = (float)m;
  rstd = rsqrtf((float)var + VPT_NORM_EPS);
}

// block-level (256 threads) reduction of two partial sums, then ONE pair of fp64 atomics per block
__device__ __forceinline__ void block_stats_atomic(float s_sum, float s_sq, double* stats_out, int f) {
  __shared__ float red_[8];
  s_sum = wave_sum(s_sum);
  s_sq = wave_sum(s_sq);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red_[w] = s_sum; red_[4 + w] = s_sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(stats_out + 2 * f, (double)((red_[0] + red_[1]) + (red_[2] + red_[3])));
    atomicAdd(stats_out + 2 * f + 1, (double)((red_[4] + red_[5]) + (red_[6] + red_[7])));
  }
}

// ---- counter-based random numbers (Philox4x32-10, Salmon et al. SC'11: the generator torch's CUDA / HIP backend uses too) ------------
// CategoricalActionHead.sample draws u = th.rand_like(logits) from torch's generator (lib/action_head.py:200); inside a captured
// acting step the draw has to be a pure function of device-resident state, so the head kernel generates its uniforms itself:
// key = the 64-bit seed, counter = (element / 4, row, step lo, step hi ^ stream << 24), word element % 4 of the block, top 24 bits
// -> u in [0, 1) exactly as torch.rand's float32 path (x >> 8) * 2^-24.  `step` is a device counter the caller advances once per
// acting step (vpt_act_epilogue); `stream` separates the heads.  oracle/philox.py restates it for the tests (bit-exact).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float vpt_philox_uniform(uint64_t seed, uint64_t step, uint32_t stream, uint32_t row, uint32_t elem) {
  uint32_t o[4];
  philox4x32_10(elem >> 2, row, (uint32_t)step, (uint32_t)(step >> 32) ^ (stream << 24), (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return (float)(o[elem & 3] >> 8) * (1.0f / 16777216.0f);
}

// streaming accesses of the HBM-bound kernels (touched once per pass; nothing on the chip can hold a 1 GB chunk)
#ifndef VPT_STREAM_PLAIN   // nontemporal by default: measured -2 ms on conv_bwd_prep, -0.5 ms on the affine backward per BC step
#define VPT_LD_STREAM(p) __builtin_nontemporal_load(p)
#define VPT_ST_STREAM(v, p) __builtin_nontemporal_store((v), (p))
#else
#define VPT_LD_STREAM(p) (*(p))
#define VPT_ST_STREAM(v, p) (*(p) = (v))
#endif

// XCD-aware bijective remap of a 1-D grid: the dispatcher places block b on XCD b%8; give every XCD a
// contiguous range of logical tiles so neighbouring tiles (shared halos / weights) share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// ---- host helper: large dynamic LDS opt-in, once per (kernel, device) ------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device attribute; a process driving several GPUs needs it on each.
static inline bool vpt_lds_optin(const void* func, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (dev < 0 || dev >= 64) return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  const unsigned long long bit = 1ull << dev;
  if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return true;
  if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
  return true;
}

