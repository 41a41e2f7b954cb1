// bf16 MFMA GEMM for every dense layer on the path (gfx950):  C[M,N] = A[M,K] * W[N,K]^T  (+bias, ReLU,
// +fp32 residual), fp32 accumulate, fp32 and/or bf16 output.
//
// Replaces nn.Linear inside FanInInitReLULayer (lib/util.py:58-82: ImpalaCNN.dense, ImgObsProcess.linear,
// mlp0, mlp1, lastlayer), the q/k/v/r/proj projections of SelfAttentionLayer (lib/xf.py:251-254,334-356;
// fused into one N = 3*hid + 10*heads GEMM), and the action/value head linears (lib/action_head.py:164,
// lib/scaled_mse_head.py:35; fused into one N = 8641+121+1 GEMM).  The preceding LayerNorm is applied by
// vpt_layernorm_kernel, which also produces this kernel's bf16 A operand.
//
// Weights are pre-packed host side as [ntile][K/32][128][32] bf16 (N zero-padded to 128) so one k-step's
// B tile is a contiguous 16 KB run.  Workgroup tile 256 x 128, k-step 64, 4 waves of 128 x 64 (4x2 MFMA
// 32x32x16 accumulators); register-staged double-step prefetch (global loads issued before the MFMAs of
// the current step, ds_write after the barrier); padded LDS rows (144 B / 80 B) keep ds_read_b128
// conflict-free.  Optional split-K with fp32 atomics for the K = 65536 dense layer.
#include "vpt_common.h"
#include "vpt_kernels.h"

#define GA_RS 144
#define GA_BYTES (256 * GA_RS)  // 36864
#define GB_RS 80
#define GB_BYTES (2 * 128 * GB_RS)  // 20480

// Epilogue variants are COMPILE-TIME (round 4): one instantiation per combination of stages a call site uses, so the kernel carries only the
// pointers it needs across the main loop (the single runtime-flag version held all of them: 94 spilled SGPRs, reloaded around every tile) and
// the epilogue has no branches.  GE_VEC: every row stride and N are multiples of 4 -- the MFMA operands are swapped (weights = A rows,
// tokens = B columns, as in vpt_conv3x3_kernel) so that a lane holds 4 CONSECUTIVE output columns of one token per accumulator group and
// every access is 16 bytes (fp32) / 8 bytes (16-bit): 32 store instructions per wave and output instead of 128 four-byte ones.
// GE_GENERIC keeps the round-3 epilogue (runtime flags, any stride) for everything else.
enum : unsigned { GE_BIAS = 1, GE_RELU = 2, GE_MASK = 4, GE_RES = 8, GE_OUTF = 16, GE_OUTB = 32, GE_SPLIT = 64, GE_VEC = 128, GE_GENERIC = 0x8000 };

// ---- vector epilogue (GE_VEC instantiations; shared by the 256 x 128 and the 256 x 256 kernel): lane = token row (l31 of the 32-row subtile),
// accumulator group g = output columns 8 g + 4 hi .. + 3 of the wave's 64-column slice that starts at col0 ----
template <unsigned F>
__device__ __forceinline__ void gemm_epilogue_vec(const VptGemmArgs& a, const f32x16 (&acc)[4][2], int row_base, int col0, int l31, int split) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = row_base + m * 32 + l31;
    const bool rvalid = row < a.M;
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = col0 + n2 * 32 + 8 * g;
        if (!(rvalid && col < a.N)) continue;        // (N % 4 == 0: a group of four columns is valid or invalid as a whole)
        f32x4 v = {acc[m][n2][4 * g + 0], acc[m][n2][4 * g + 1], acc[m][n2][4 * g + 2], acc[m][n2][4 * g + 3]};
        if constexpr ((F & GE_SPLIT) != 0) {
          *(f32x4*)(a.out_f32 + ((size_t)split * a.M + row) * a.ldc + col) = v;
          continue;
        }
        if constexpr ((F & GE_BIAS) != 0) v += *(const f32x4*)(a.bias + col);
        if constexpr ((F & GE_RELU) != 0) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
        if constexpr ((F & GE_MASK) != 0) {          // ReLU backward: gate by the saved activation
          const u32x2 mk = *(const u32x2*)(a.mask + (size_t)row * a.ldm + col);
          if (!(op16_lo_to_f32(mk.x) > 0.f)) v.x = 0.f;
          if (!(op16_hi_to_f32(mk.x) > 0.f)) v.y = 0.f;
          if (!(op16_lo_to_f32(mk.y) > 0.f)) v.z = 0.f;
          if (!(op16_hi_to_f32(mk.y) > 0.f)) v.w = 0.f;
        }
        if constexpr ((F & GE_RES) != 0) v += *(const f32x4*)(a.res + (size_t)row * a.ldr + col);
        if constexpr ((F & GE_OUTF) != 0) *(f32x4*)(a.out_f32 + (size_t)row * a.ldc + col) = v;
        if constexpr ((F & GE_OUTB) != 0) {
          const u32x2 pk = {pack_op16x2(v.x, v.y), pack_op16x2(v.z, v.w)};
          *(u32x2*)(a.out_bf16 + (size_t)row * a.ldcb + col) = pk;
        }
      }
    }
  }
}

template <unsigned F>
__global__ __launch_bounds__(256, 2) void vpt_gemm_kernel(VptGemmArgs a) {
  constexpr bool VEC = (F & GE_VEC) != 0 && F != GE_GENERIC;
  __shared__ __attribute__((aligned(16))) unsigned char smem[GA_BYTES + GB_BYTES];
  // GE_GENERIC: the epilogue's pointers and strides wait in LDS while the main loop runs (held in SGPRs they did not fit: 94 spills)
  __shared__ VptGemmArgs epi_args;
  if (F == GE_GENERIC && threadIdx.x == 0) epi_args = a;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int mtiles = (a.M + 255) >> 8, NT = (a.N + 127) >> 7;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = L % NT; L /= NT;
  const int mt = L % mtiles;
  const int split = L / mtiles;
  const int m0 = mt * 256;
  const int nsteps_total = a.K >> 6;
  const int per = (nsteps_total + a.splitk - 1) / a.splitk;
  const int s_begin = split * per;
  const int s_end = min(s_begin + per, nsteps_total);
  if (s_begin >= s_end) return;

  // staging maps: A row (m0 + arow + 32 m), 16-byte part apart.  Rows beyond M re-read row M - 1 (clamped offset, no branch and
  // no select in the main loop): an output row depends on its own A row only, and the epilogue never stores rows >= M.
  const int arow = tid >> 3, apart = tid & 7;
  const char* abase = (const char*)(a.A + (size_t)m0 * a.lda);           // wave-uniform
  unsigned aoff[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) aoff[m] = (unsigned)min(arow + 32 * m, a.M - 1 - m0) * (unsigned)a.lda * 2u + apart * 16u;
#define A_LOAD(m_, s_) (*(const u32x4*)(abase + (size_t)(s_) * 128 + aoff[m_]))
  unsigned char* ast = smem + arow * GA_RS + apart * 16;
  // B staging: a ds_write_b128 is served in groups of 8 consecutive lanes with banks taken modulo 32, and two 64-byte rows
  // one pitch (80 B) apart overlap in four banks -- the lanes of a group take rows R and R + 4 instead (320 B = 16 banks apart)
  const int brow = ((tid >> 5) << 3) + ((tid >> 3) & 3) + ((tid >> 2) & 1) * 4;
  const vpt_op16* wbase = a.wpk + (size_t)nt * (a.K >> 5) * 4096 + (brow * 4 + (tid & 3)) * 8;
  unsigned char* bst = smem + GA_BYTES + brow * GB_RS + (tid & 3) * 16;

  u32x4 areg[8], breg[4];
#pragma unroll
  for (int m = 0; m < 8; ++m) areg[m] = A_LOAD(m, s_begin);
#pragma unroll
  for (int m = 0; m < 4; ++m) breg[m] = *(const u32x4*)(wbase + (size_t)s_begin * 8192 + m * 2048);
#pragma unroll
  for (int m = 0; m < 8; ++m) *(u32x4*)(ast + m * 32 * GA_RS) = areg[m];
#pragma unroll
  for (int m = 0; m < 4; ++m) *(u32x4*)(bst + m * 64 * GB_RS) = breg[m];
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const unsigned char* aL = smem + (wm * 128 + l31) * GA_RS + hi * 16;
  const unsigned char* bL = smem + GA_BYTES + (wn * 64 + l31) * GB_RS + hi * 16;

  // Main loop, one k-step (64) = 4 slices of 16 = 4 groups of 8 MFMAs per wave.  The schedule is explicit and pinned with
  // sched_barrier, the pattern of vpt_conv3x3_kernel (round 2), within this kernel's register budget (128 accumulators + 48
  // staging registers): the four A fragments ROLL -- fragment m of slice kk + 1 is requested into its own registers right after
  // the two MFMAs that consumed fragment m of slice kk, six or more MFMAs ahead of its use -- and only the two B fragments, which
  // all eight MFMAs of a slice read, are double-buffered; the next step's twelve global loads ride in the remaining slots; the
  // barrier that frees the LDS tile stands in front of the LAST slice's MFMAs (its fragments are in registers), so the ds_writes
  // of the next tile issue between those MFMAs.  The first version read six fragments, waited, issued eight MFMAs, four times
  // per step, plus two bare barriers: 670 TF/s on the trunk GEMMs.
  op16x8 fa[4], fb[2][2];
#define GSB() __builtin_amdgcn_sched_barrier(0)
#define GMM(setb_, m_, n_) acc[m_][n_] = VEC ? VPT_MFMA_32X32X16(fb[setb_][n_], fa[m_], acc[m_][n_], 0, 0, 0) : VPT_MFMA_32X32X16(fa[m_], fb[setb_][n_], acc[m_][n_], 0, 0, 0)
#define GFA(kk_, m_) fa[m_] = *(const op16x8*)(aL + (m_) * (32 * GA_RS) + (kk_) * 32)
#define GFB(setb_, kk_, n_) fb[setb_][n_] = *(const op16x8*)(bL + ((kk_) >> 1) * (128 * GB_RS) + (n_) * (32 * GB_RS) + ((kk_) & 1) * 32)
#define GLA(m_) areg[m_] = A_LOAD(m_, s + 1)
#define GLB(m_) breg[m_] = *(const u32x4*)(wbase + (size_t)(s + 1) * 8192 + (m_) * 2048)
#define GNOP() ((void)0)
#define GGROUP(setb_, nkk_, X0, X1, X2, X3)                                                               \
  do {                                                                                                    \
    GMM(setb_, 0, 0); GFB(1 - (setb_), nkk_, 0); GFB(1 - (setb_), nkk_, 1); GSB();                        \
    GMM(setb_, 0, 1); GFA(nkk_, 0); GSB();                                                                \
    GMM(setb_, 1, 0); X0; GSB();                                                                          \
    GMM(setb_, 1, 1); GFA(nkk_, 1); X1; GSB();                                                            \
    GMM(setb_, 2, 0); X2; GSB();                                                                          \
    GMM(setb_, 2, 1); GFA(nkk_, 2); X3; GSB();                                                            \
    GMM(setb_, 3, 0); GSB();                                                                              \
    GMM(setb_, 3, 1); GFA(nkk_, 3); GSB();                                                                \
  } while (0)
#define GFIRST() do { GFB(0, 0, 0); GFA(0, 0); GFB(0, 0, 1); GFA(0, 1); GFA(0, 2); GFA(0, 3); GSB(); } while (0)
  GFIRST();
#pragma unroll 1
  for (int s = s_begin; s + 1 < s_end; ++s) {   // every step but the last: prefetches its successor
    GGROUP(0, 1, GLB(0), GLB(1), GLB(2), GLB(3));
    GGROUP(1, 2, GLA(0), GLA(1), GLA(2), GLA(3));
    GGROUP(0, 3, GLA(4), GLA(5), GLA(6), GLA(7));
    __syncthreads();   // every wave holds its last fragments of this step: the tile may be overwritten
    GSB();
    // last slice's MFMAs with the next tile's ds_writes between them (each waits for its own global load)
    GMM(1, 0, 0); *(u32x4*)(bst + 0 * 64 * GB_RS) = breg[0]; *(u32x4*)(bst + 1 * 64 * GB_RS) = breg[1]; GSB();
    GMM(1, 0, 1); *(u32x4*)(bst + 2 * 64 * GB_RS) = breg[2]; *(u32x4*)(bst + 3 * 64 * GB_RS) = breg[3]; GSB();
    GMM(1, 1, 0); *(u32x4*)(ast + 0 * 32 * GA_RS) = areg[0]; *(u32x4*)(ast + 1 * 32 * GA_RS) = areg[1]; GSB();
    GMM(1, 1, 1); *(u32x4*)(ast + 2 * 32 * GA_RS) = areg[2]; *(u32x4*)(ast + 3 * 32 * GA_RS) = areg[3]; GSB();
    GMM(1, 2, 0); *(u32x4*)(ast + 4 * 32 * GA_RS) = areg[4]; *(u32x4*)(ast + 5 * 32 * GA_RS) = areg[5]; GSB();
    GMM(1, 2, 1); *(u32x4*)(ast + 6 * 32 * GA_RS) = areg[6]; *(u32x4*)(ast + 7 * 32 * GA_RS) = areg[7]; GSB();
    GMM(1, 3, 0); GSB();
    GMM(1, 3, 1); GSB();
    __syncthreads();
    GSB();
    GFIRST();
  }
  {   // last step
    GGROUP(0, 1, GNOP(), GNOP(), GNOP(), GNOP());
    GGROUP(1, 2, GNOP(), GNOP(), GNOP(), GNOP());
    GGROUP(0, 3, GNOP(), GNOP(), GNOP(), GNOP());
#pragma unroll
    for (int m = 0; m < 4; ++m) { GMM(1, m, 0); GMM(1, m, 1); }
  }
#undef GFIRST
#undef GGROUP
#undef GNOP
#undef GLA
#undef A_LOAD
#undef GLB
#undef GFA
#undef GFB
#undef GMM
#undef GSB

  if constexpr (VEC) {
    gemm_epilogue_vec<F>(a, acc, m0 + wm * 128, nt * 128 + wn * 64 + 4 * hi, l31, split);
    return;
  }
  // ---- generic epilogue (direct from the accumulator layout: lane = column, 16 rows per accumulator) ----
  {
    const volatile VptGemmArgs* ev = &epi_args;       // (written before the main loop's first barrier)
    a.bias = ev->bias; a.res = ev->res; a.out_f32 = ev->out_f32; a.out_bf16 = ev->out_bf16; a.mask = ev->mask;
    a.ldr = ev->ldr; a.ldc = ev->ldc; a.ldcb = ev->ldcb; a.ldm = ev->ldm; a.relu = ev->relu; a.atomic_out = ev->atomic_out;
  }
  // One accumulator (16 values) at a time, the optional stages as separate uniform-branch loops over 32-bit offsets
  // from a per-tile base: the straightforward per-element chain of pointer tests and 64-bit index products made the
  // compiler spill accumulators and cost ~100 us per tile (an intercept of 0.4 ms on an 8192 x 8192 output).
  const int rbase = m0 + wm * 128 + 4 * hi;
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) {
    const int col = nt * 128 + wn * 64 + n2 * 32 + l31;
    const bool cvalid = col < a.N;
    const float bv = (a.bias && split == 0 && cvalid) ? a.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row0 = rbase + m * 32;
      float v[16];
      // (the validity test is re-evaluated at every use: sixteen stored lane masks were 32 SGPRs, spilled together with everything else)
#define OK_(r_) (cvalid && (row0 + ((r_) & 3) + 8 * ((r_) >> 2)) < a.M)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[m][n2][r] + bv;
      if (a.atomic_out) {   // split-K: this split's partial product goes to its own [M][ldc] slice (deterministic; the caller sums)
        float* o = a.out_f32 + ((size_t)split * a.M + row0) * a.ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (OK_(r)) o[((r & 3) + 8 * (r >> 2)) * a.ldc] = v[r];
        continue;
      }
      if (a.relu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.mask) {         // ReLU backward: gate by the saved activation
        const vpt_op16* mk = a.mask + (size_t)row0 * a.ldm + col;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (OK_(r) && !((float)mk[((r & 3) + 8 * (r >> 2)) * a.ldm] > 0.f)) v[r] = 0.f;
      }
      if (a.res) {
        const float* rp = a.res + (size_t)row0 * a.ldr + col;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (OK_(r)) v[r] += rp[((r & 3) + 8 * (r >> 2)) * a.ldr];
      }
      if (a.out_f32) {
        float* o = a.out_f32 + (size_t)row0 * a.ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (OK_(r)) o[((r & 3) + 8 * (r >> 2)) * a.ldc] = v[r];
      }
      if (a.out_bf16) {
        vpt_op16* o = a.out_bf16 + (size_t)row0 * a.ldcb + col;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (OK_(r)) o[((r & 3) + 8 * (r >> 2)) * a.ldcb] = (vpt_op16)v[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same GEMM on a 256 x 256 tile with EIGHT waves (2 x 4 of 128 x 64), one workgroup per CU (round 4).  The 256 x 128 kernel above stages
// both operands through registers into ONE padded LDS tile: 12 global loads + 12 ds_write_b128 per thread and two barriers per 32 MFMAs.  Here
// both operands of a k-step arrive by LDS-DMA (global_load_lds, 8 x 16 bytes per thread) into a DOUBLE buffer, one step ahead: no staging
// registers, no LDS write instructions, one barrier per step (in front of the step's last eight MFMAs, as in vpt_conv3x3_kernel: every wave
// then holds its last fragments, the next step's tile has landed, its first fragments are requested behind the barrier and arrive under those
// MFMAs), a third less L2 -> LDS traffic per FLOP.  LDS image of a step: A [2 x 32 k][256 rows][64 B], B likewise (the packed weights' own
// 128-row blocks, two of them) -- unpadded; the DMA is lane-linear, so the XOR swizzle of the four 16-byte pieces of a row by (row >> 2) & 3
// is applied on the SOURCE address and undone in the fragment address (conflict-free ds_read_b128, the conv kernel's weight-tile scheme).
// Same K order and MFMA sequence per output element as the 256 x 128 kernel: bit-identical results (tests/test_gpu_kernels.py).
#define G2_HALF 32768                 // one operand of one step: 2 x 256 rows x 64 B
#define G2_BUF (2 * G2_HALF)          // A + B of a step
template <unsigned F>
__global__ __launch_bounds__(512, 2) void vpt_gemm256_kernel(VptGemmArgs a) {
  static_assert((F & GE_VEC) != 0 && (F & GE_SPLIT) == 0, "vector epilogue, no split-K");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G2_BUF];   // 128 KB
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int hi = lane >> 5, l31 = lane & 31;
  const int NT2 = (a.N + 255) >> 8, NT128 = (a.N + 127) >> 7;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = L % NT2;
  const int mt = L / NT2;
  const int m0 = mt * 256;
  const int nsteps = a.K >> 6;

  // ---- DMA source addresses: chunk idx = tid + 512 j  ->  32-k block kb = j >> 1, row r = (tid >> 2) + 128 (j & 1), LDS piece p = tid & 3,
  // which holds the row's source piece p ^ ((r >> 2) & 3) ----
  const int r0 = tid >> 2, psrc = (tid & 3) ^ ((r0 >> 2) & 3);        // (r0 + 128 has the same swizzle key)
  const char* abase = (const char*)(a.A + (size_t)m0 * a.lda);         // wave-uniform
  unsigned aoff[2];
#pragma unroll
  for (int jr = 0; jr < 2; ++jr) aoff[jr] = (unsigned)min(r0 + 128 * jr, a.M - 1 - m0) * (unsigned)a.lda * 2u + psrc * 16u;   // rows beyond M re-read row M - 1
  const char* wsrc[2];
#pragma unroll
  for (int jr = 0; jr < 2; ++jr)
    wsrc[jr] = (const char*)(a.wpk + (size_t)min(2 * nt + jr, NT128 - 1) * (a.K >> 5) * 4096 + r0 * 32 + psrc * 8);   // (an odd number of 128-column tiles: the missing half re-reads the last one)
  unsigned char* const dma_dst = smem + w * 1024;
#define G2_ISSUE(s_, buf_, j_)                                                                             \
  do {                                                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + (size_t)(s_) * 128 + ((j_) >> 1) * 64 + aoff[(j_) & 1]), \
                                     (__attribute__((address_space(3))) void*)(dma_dst + (buf_) + (j_) * 8192), 16, 0, 0);                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[(j_) & 1] + ((size_t)(s_) * 2 + ((j_) >> 1)) * 8192), \
                                     (__attribute__((address_space(3))) void*)(dma_dst + (buf_) + G2_HALF + (j_) * 8192), 16, 0, 0);           \
  } while (0)

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // fragment base addresses: slice kk = (32-k block kk >> 1, 16-k half kk & 1); piece ((kk & 1) * 2 + hi) ^ swizzle key of the lane's row
  const int sw = (l31 >> 2) & 3;
  // [buffer][16-k half]: the step's buffer is chosen by swapping these per-lane addresses (the ds_read offset field has 16 bits)
  const unsigned char* aC[2] = {smem + (wm * 128 + l31) * 64 + (((0 + hi) ^ sw) << 4), smem + (wm * 128 + l31) * 64 + (((2 + hi) ^ sw) << 4)};
  const unsigned char* bC[2] = {smem + G2_HALF + (wn * 64 + l31) * 64 + (((0 + hi) ^ sw) << 4), smem + G2_HALF + (wn * 64 + l31) * 64 + (((2 + hi) ^ sw) << 4)};
  const unsigned char* aN[2] = {aC[0] + G2_BUF, aC[1] + G2_BUF};
  const unsigned char* bN[2] = {bC[0] + G2_BUF, bC[1] + G2_BUF};

  op16x8 fa[4], fb[2][2];
#define QSB() __builtin_amdgcn_sched_barrier(0)
#define QMM(setb_, m_, n_) acc[m_][n_] = VPT_MFMA_32X32X16(fb[setb_][n_], fa[m_], acc[m_][n_], 0, 0, 0)
#define QFA(NXT_, kk_, m_) fa[m_] = *(const op16x8*)(((NXT_) ? aN : aC)[(kk_) & 1] + ((kk_) >> 1) * 16384 + (m_) * 2048)
#define QFB(NXT_, setb_, kk_, n_) fb[setb_][n_] = *(const op16x8*)(((NXT_) ? bN : bC)[(kk_) & 1] + ((kk_) >> 1) * 16384 + (n_) * 2048)
#define QNOP() ((void)0)
  // eight MFMAs of fragment set (fa, fb[setb_]); the fragments of slice nkk_ (buffer offset nboff_) follow into fa (rolling: fragment m right
  // behind the two MFMAs that consumed it) and fb[1 - setb_]; X0 .. X3: the step's DMA issue slots
#define QGROUP(setb_, nboff_, nkk_, X0, X1, X2, X3)                                                        \
  do {                                                                                                     \
    QMM(setb_, 0, 0); QFB(nboff_, 1 - (setb_), nkk_, 0); QFB(nboff_, 1 - (setb_), nkk_, 1); QSB();         \
    QMM(setb_, 0, 1); QFA(nboff_, nkk_, 0); QSB();                                                         \
    QMM(setb_, 1, 0); X0; QSB();                                                                           \
    QMM(setb_, 1, 1); QFA(nboff_, nkk_, 1); X1; QSB();                                                     \
    QMM(setb_, 2, 0); X2; QSB();                                                                           \
    QMM(setb_, 2, 1); QFA(nboff_, nkk_, 2); X3; QSB();                                                     \
    QMM(setb_, 3, 0); QSB();                                                                               \
    QMM(setb_, 3, 1); QFA(nboff_, nkk_, 3); QSB();                                                         \
  } while (0)
#define QWAIT_BARRIER()                                                                                    \
  do {                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("" ::: "memory");                                                                         \
    QSB();                                                                                                 \
  } while (0)
  // one k-step out of the current buffer; MORE_: a step follows (its tile is requested here into the other buffer, byte offset nbuf_, and its
  // first fragments are read at the end)
#define QSTEP(s_, nbuf_, MORE_)                                                                            \
  do {                                                                                                     \
    if (MORE_) {                                                                                           \
      QGROUP(0, 0, 1, G2_ISSUE((s_) + 1, nbuf_, 0), QNOP(), G2_ISSUE((s_) + 1, nbuf_, 1), QNOP());          \
      QGROUP(1, 0, 2, G2_ISSUE((s_) + 1, nbuf_, 2), QNOP(), G2_ISSUE((s_) + 1, nbuf_, 3), QNOP());          \
      QGROUP(0, 0, 3, QNOP(), QNOP(), QNOP(), QNOP());                                                     \
      QWAIT_BARRIER();   /* every wave holds its last fragments; the next tile has landed */                \
      QGROUP(1, 1, 0, QNOP(), QNOP(), QNOP(), QNOP());                                                     \
    } else {                                                                                               \
      QGROUP(0, 0, 1, QNOP(), QNOP(), QNOP(), QNOP());                                                     \
      QGROUP(1, 0, 2, QNOP(), QNOP(), QNOP(), QNOP());                                                     \
      QGROUP(0, 0, 3, QNOP(), QNOP(), QNOP(), QNOP());                                                     \
      _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) { QMM(1, m_, 0); QMM(1, m_, 1); }                    \
    }                                                                                                      \
  } while (0)

  // prologue: tile 0
#pragma unroll
  for (int j = 0; j < 4; ++j) G2_ISSUE(0, 0, j);
  QWAIT_BARRIER();
  QFB(0, 0, 0, 0); QFA(0, 0, 0); QFB(0, 0, 0, 1); QFA(0, 0, 1); QFA(0, 0, 2); QFA(0, 0, 3);
  QSB();
#pragma unroll 1
  for (int s = 0; s + 1 < nsteps; ++s) {
    const int nbuf = ((s + 1) & 1) * G2_BUF;     // wave-uniform: the DMA's LDS base goes through M0
    QSTEP(s, nbuf, true);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                // the next step's buffer becomes the current one
      const unsigned char* t = aC[h]; aC[h] = aN[h]; aN[h] = t;
      t = bC[h]; bC[h] = bN[h]; bN[h] = t;
    }
  }
  QSTEP(nsteps - 1, 0, false);
#undef QSTEP
#undef QWAIT_BARRIER
#undef QGROUP
#undef QNOP
#undef QFB
#undef QFA
#undef QMM
#undef QSB
#undef G2_ISSUE
  gemm_epilogue_vec<F>(a, acc, m0 + wm * 128, nt * 256 + wn * 64 + 4 * hi, l31, 0);
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient GEMM in the "TN" form:  C[n1][n2] (+)= sum_m A[m][n1] * B[m][n2]  with A, B bf16 ROW-major over the
// reduction index m (frames / tokens) -- exactly how dY [M][N] and X [M][K] lie in memory, so dW = dY^T X needs no
// transposed copies of the activations.  The k-contiguous MFMA fragments come from gfx950's LDS transpose read
// (ds_read_b64_tr_b16: 16 lanes read a [4 m][16 n] block of the row-major tile, each receives one column's 4 m's), the
// trick of vpt_conv_wgrad.hip.  Tile 256 (n1) x 128 (n2), k-step 64 m-rows, 4 waves of 128 x 64, register-staged
// prefetch; LDS row pitches 576 B / 320 B (= 16 banks mod 64: the four rows of a transpose read land on disjoint banks).
#define TA_RS 576
#define TA_BYTES (64 * TA_RS)   // 36864
#define TB_RS 320
#define TB_BYTES (64 * TB_RS)   // 20480
__device__ __forceinline__ op16x8 tn_frag(const unsigned char* p, int row_stride) {   // 8 consecutive m of this lane's column
  const op16x4 a = lds_tr16_read(p);
  const op16x4 b = lds_tr16_read(p + 4 * row_stride);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 2) void vpt_gemm_tn_kernel(VptGemmTnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TA_BYTES + TB_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int t2 = (a.N2 + 127) >> 7;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int j2 = L % t2; L /= t2;
  const int j1 = L;
  const int c1 = j1 * 256, c2 = j2 * 128;
  const int nsteps = (a.M + 63) >> 6;

  // staging: A tile = 64 rows x 512 B = 2048 chunks of 16 B (8 per thread), B tile = 64 x 256 B = 1024 chunks (4 per thread)
  u32x4 areg[8], breg[4];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load = [&](int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = tid + 256 * i, r = q >> 5, c = (q & 31) * 8;
      const int m = s * 64 + r, col = c1 + c;
      areg[i] = (m < a.M && col < a.N1) ? *(const u32x4*)(a.A + (size_t)m * a.lda + col) : zero4;   // N1, N2 multiples of 8
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i, r = q >> 4, c = (q & 15) * 8;
      const int m = s * 64 + r, col = c2 + c;
      breg[i] = (m < a.M && col < a.N2) ? *(const u32x4*)(a.B + (size_t)m * a.ldb + col) : zero4;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = tid + 256 * i;
      *(u32x4*)(smem + (q >> 5) * TA_RS + (q & 31) * 16) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;
      *(u32x4*)(smem + TA_BYTES + (q >> 4) * TB_RS + (q & 15) * 16) = breg[i];
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  // per-lane part of a fragment address: 16-lane group g reads m-rows 8*(g>>1) + (i>>2) (+4), columns 16*(g&1) + 4*(i&3)..+3
  const int g16 = lane >> 4, i16 = lane & 15;
  const int colb = (16 * (g16 & 1) + 4 * (i16 & 3)) * 2, rowl = 8 * (g16 >> 1) + (i16 >> 2);
  const unsigned char* aL = smem + rowl * TA_RS + (wm * 128) * 2 + colb;
  const unsigned char* bL = smem + TA_BYTES + rowl * TB_RS + (wn * 64) * 2 + colb;

  load(0);
  store();
  __syncthreads();
  // Round 5: while the eight MFMAs of 16-row slice kk issue, the six fragments of slice kk + 1 are requested, each behind the last MFMA
  // that reads the register it lands in, pinned with sched_barrier (left alone the scheduler sinks every transpose read next to its use and the
  // matrix pipe drains for an LDS round trip per slice -- the finding of vpt_conv3x3_kernel round 2 and vpt_conv_wgrad_kernel; the first
  // version of this loop read all six fragments, waited, then issued its MFMAs: 645 TF/s).  Only slice 0 of a step -- its tile is stored behind
  // the barrier -- is exposed.
  op16x8 af[4], bfr[2];         // ONE fragment set that rolls: a fragment of the next slice is requested behind the last MFMA that used its register
#define TN_SB() __builtin_amdgcn_sched_barrier(0)
#define TN_LDA(kk_, m_) af[m_] = tn_frag(aL + (kk_) * 16 * TA_RS + (m_) * 64, TA_RS)
#define TN_LDB(kk_, n_) bfr[n_] = tn_frag(bL + (kk_) * 16 * TB_RS + (n_) * 64, TB_RS)
#define TN_MM(m_, n_) acc[m_][n_] = VPT_MFMA_32X32X16(af[m_], bfr[n_], acc[m_][n_], 0, 0, 0)
#define TN_SLICE(kk_, NEXT)                                                                               \
  do {                                                                                                    \
    TN_MM(0, 0); TN_SB();                                                                                 \
    TN_MM(1, 0); TN_SB();                                                                                 \
    TN_MM(2, 0); TN_SB();                                                                                 \
    TN_MM(3, 0); if (NEXT) TN_LDB((kk_) + 1, 0); TN_SB();                                                 \
    TN_MM(0, 1); if (NEXT) TN_LDA((kk_) + 1, 0); TN_SB();                                                 \
    TN_MM(1, 1); if (NEXT) TN_LDA((kk_) + 1, 1); TN_SB();                                                 \
    TN_MM(2, 1); if (NEXT) TN_LDA((kk_) + 1, 2); TN_SB();                                                 \
    TN_MM(3, 1); if (NEXT) { TN_LDA((kk_) + 1, 3); TN_LDB((kk_) + 1, 1); } TN_SB();                       \
  } while (0)
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) load(s + 1);
    TN_LDA(0, 0); TN_LDB(0, 0); TN_LDA(0, 1); TN_LDA(0, 2); TN_LDA(0, 3); TN_LDB(0, 1);
    TN_SB();
    TN_SLICE(0, true);
    TN_SLICE(1, true);
    TN_SLICE(2, true);
    TN_SLICE(3, false);
    __syncthreads();
    if (more) store();
    __syncthreads();
  }
#undef TN_SLICE
#undef TN_MM
#undef TN_LDB
#undef TN_LDA
#undef TN_SB
  // epilogue: lane = column (n2), 16 rows (n1) per accumulator; optional accumulate into the existing output
  const int rbase = c1 + wm * 128 + 4 * hi;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = c2 + wn * 64 + n * 32 + l31;
    if (col >= a.N2) continue;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float* o = a.C + (size_t)(rbase + m * 32) * a.ldc + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (rbase + m * 32 + dr < a.N1) o[dr * a.ldc] = a.accumulate ? o[dr * a.ldc] + acc[m][n][r] : acc[m][n][r];
      }
    }
  }
}

extern "C" int vpt_gemm_tn_launch(const VptGemmTnArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->N1 <= 0 || a->N2 <= 0 || (a->N1 & 7) || (a->N2 & 7) || (a->lda & 7) || (a->ldb & 7)) return -1;
  const long grid = (long)((a->N1 + 255) >> 8) * ((a->N2 + 127) >> 7);
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_gemm_tn_kernel, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Second stage of a split-K linear: out = epilogue( sum_s part[s] ) with the GEMM's full epilogue (bias, ReLU, gate mask,
// fp32 residual, fp32 / bf16 outputs).  Fixed summation order -> deterministic.  Used for mid-size M (e.g. one
// 128-frame IDM window), where the 256 x 128 tiling alone would put only N/128 workgroups on 256 CUs.
__global__ __launch_bounds__(256) void vpt_splitk_epilogue_kernel(const float* __restrict__ part, int splitk, VptGemmArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.M * a.N;
  if (i >= total) return;
  const int row = (int)(i / a.N), col = (int)(i - (long)row * a.N);
  float v = 0.f;
  for (int sp = 0; sp < splitk; ++sp) v += part[(size_t)sp * total + i];
  if (a.bias) v += a.bias[col];
  if (a.relu) v = fmaxf(v, 0.f);
  if (a.mask && !((float)a.mask[(size_t)row * a.ldm + col] > 0.f)) v = 0.f;
  if (a.res) v += a.res[(size_t)row * a.ldr + col];
  if (a.out_f32) a.out_f32[(size_t)row * a.ldc + col] = v;
  if (a.out_bf16) a.out_bf16[(size_t)row * a.ldcb + col] = (vpt_op16)v;
}

extern "C" int vpt_splitk_epilogue_launch(const float* part, int splitk, const VptGemmArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->N <= 0 || splitk < 1 || !part) return -1;
  const long total = (long)a->M * a->N;
  hipLaunchKernelGGL(vpt_splitk_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, part, splitk, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Second stage of the split-K dense layer with ImpalaCNN.dense's LayerNorm(65536) FOLDED in (lib/impala_cnn.py:177-194: LN -> Linear -> the
// ReLU comes later): the GEMM ran on the raw block output x with weights op16(W * gain), so per frame f
//     out[f][n] = rstd_f * sum_s part[s][f][n]  -  rstd_f * mean_f * sg[n]  +  sb[n],     sg[n] = sum_k op16(W g)[n][k],  sb[n] = sum_k W[n][k] bias[k]
// with (mean_f, rstd_f) from the producer's frame statistics -- the per-element affine pass over the 16 x 16 x C tensor disappears.
// Fixed summation order over the splits: deterministic.
__global__ __launch_bounds__(256) void vpt_dense_fold_epilogue_kernel(const float* __restrict__ part, int splitk, const double* __restrict__ stats, double inv_count,
                                                                      const float* __restrict__ sg, const float* __restrict__ sb, float* __restrict__ out, int M, int N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)M * N;
  if (i >= total) return;
  const int row = (int)(i / N), col = (int)(i - (long)row * N);
  float mean, rstd;
  frame_mean_rstd(stats, row, inv_count, mean, rstd);
  float v = 0.f;
  for (int sp = 0; sp < splitk; ++sp) v += part[(size_t)sp * total + i];
  out[i] = fmaf(rstd, v, fmaf(-rstd * mean, sg[col], sb[col]));
}

extern "C" int vpt_dense_fold_epilogue_launch(const float* part, int splitk, const double* stats, double inv_count, const float* sg, const float* sb, float* out,
                                              int M, int N, hipStream_t stream) {
  if (M <= 0 || N <= 0 || splitk < 1 || !part || !stats || !sg || !sb || !out) return -1;
  const long total = (long)M * N;
  hipLaunchKernelGGL(vpt_dense_fold_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, part, splitk, stats, inv_count, sg, sb, out, M, N);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int vpt_gemv_launch(const VptGemmArgs* a, hipStream_t stream);

extern "C" int vpt_gemm_launch(const VptGemmArgs* a, hipStream_t stream) {
  if (a->tiling == 2 && a->M > 8) return -1;
  if (a->M > 0 && a->M <= 8 && (a->tiling == 0 || a->tiling == 2)) return vpt_gemv_launch(a, stream);   // acting path (T = 1): HBM-bound weight stream, vpt_gemv.hip
  if (a->M <= 0 || a->N <= 0 || (a->K & 63) || a->splitk < 1 || (a->lda & 7)) return -1;
  if (a->splitk > 1 && (!a->atomic_out || a->relu || a->res || a->out_bf16 || a->mask)) return -1;
  const long grid = (long)((a->M + 255) >> 8) * ((a->N + 127) >> 7) * a->splitk;
  if (grid > 0x7fffffffL) return -2;
  unsigned f = 0;
  if (a->atomic_out) f = GE_SPLIT | GE_OUTF;
  else f = (a->bias ? GE_BIAS : 0u) | (a->relu ? GE_RELU : 0u) | (a->mask ? GE_MASK : 0u) | (a->res ? GE_RES : 0u) | (a->out_f32 ? GE_OUTF : 0u) | (a->out_bf16 ? GE_OUTB : 0u);
  const bool aligned = !(a->N & 3) && (!a->out_f32 || !(a->ldc & 3)) && (!a->out_bf16 || !(a->ldcb & 3)) && (!a->res || !(a->ldr & 3)) && (!a->mask || !(a->ldm & 3))
                       && (!a->bias || !((uintptr_t)a->bias & 15)) && (!a->out_f32 || !((uintptr_t)a->out_f32 & 15)) && (!a->res || !((uintptr_t)a->res & 15))
                       && (!a->out_bf16 || !((uintptr_t)a->out_bf16 & 7)) && (!a->mask || !((uintptr_t)a->mask & 7));
  // tiling 4: the 256 x 256 / eight-wave kernel where its grid fills the chip (one workgroup per CU: >= 192 tiles).  Bit-identical results (same K
  // order per output element).  Measured (profiles/r04_experiments.md sections 11, 12): +7-13 % per shape in a warm back-to-back loop, but -0.13 ms
  // per forward step inside the engine, where every GEMM runs once per step on cold weights and one workgroup per CU exposes its DMA prologue and
  // its 256 KB of epilogue stores per tile -- so it is NOT the default; kept selectable and under test.
  const long grid2 = (long)((a->M + 255) >> 8) * ((a->N + 255) >> 8);
  const bool big = aligned && a->splitk == 1 && !a->atomic_out && a->tiling == 4 && grid2 >= 192 && !((uintptr_t)a->A & 15);
#define GE_LAUNCH(F_)                                                                                                             \
  do {                                                                                                                            \
    if constexpr (((F_) & GE_VEC) != 0 && ((F_) & GE_SPLIT) == 0) {                                                                \
      if (big) { hipLaunchKernelGGL((vpt_gemm256_kernel<(F_)>), dim3((unsigned)grid2), dim3(512), 0, stream, *a); break; }        \
    }                                                                                                                             \
    hipLaunchKernelGGL((vpt_gemm_kernel<(F_)>), dim3((unsigned)grid), dim3(256), 0, stream, *a);                                  \
  } while (0)
  bool done = false;
  if (aligned) {
    done = true;
    switch (f) {   // the combinations the engine / trainer use (forward: qkvr / heads, proj / mlp1, mlp0, img linear, lastlayer, dense split-K; dgrad)
      case GE_BIAS | GE_OUTF: GE_LAUNCH(GE_VEC | GE_BIAS | GE_OUTF); break;
      case GE_BIAS | GE_RES | GE_OUTF: GE_LAUNCH(GE_VEC | GE_BIAS | GE_RES | GE_OUTF); break;
      case GE_RELU | GE_OUTB: GE_LAUNCH(GE_VEC | GE_RELU | GE_OUTB); break;
      case GE_RELU | GE_OUTF: GE_LAUNCH(GE_VEC | GE_RELU | GE_OUTF); break;
      case GE_RELU | GE_OUTF | GE_OUTB: GE_LAUNCH(GE_VEC | GE_RELU | GE_OUTF | GE_OUTB); break;
      case GE_OUTF: GE_LAUNCH(GE_VEC | GE_OUTF); break;
      case GE_OUTB: GE_LAUNCH(GE_VEC | GE_OUTB); break;
      case GE_RES | GE_OUTF: GE_LAUNCH(GE_VEC | GE_RES | GE_OUTF); break;
      case GE_MASK | GE_OUTB: GE_LAUNCH(GE_VEC | GE_MASK | GE_OUTB); break;
      case GE_SPLIT | GE_OUTF: GE_LAUNCH(GE_VEC | GE_SPLIT | GE_OUTF); break;
      default: done = false;
    }
  }
  if (!done) GE_LAUNCH(GE_GENERIC);
#undef GE_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
