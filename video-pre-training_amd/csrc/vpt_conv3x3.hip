// 3x3 / pad-1 convolution of the IMPALA residual CNN as an implicit GEMM on bf16 MFMA (gfx950).
//
// Replaces, per call:  FanInInitReLULayer.forward (lib/util.py:75-82) with GroupNorm(1,C) -> Conv2d(3x3,
// pad 1, no bias) -> ReLU, as used by CnnBasicBlock.conv0/conv1 (lib/impala_cnn.py:30-52) and by the
// firstconv of stacks 1..2 (lib/impala_cnn.py:86-97), plus the residual add of CnnBasicBlock.forward.
//
// Layout in HBM: activations are channel-blocked NHWC, [frame][C/32][H][W][32] bf16, so that the 18-pixel
// halo row of one 32-channel block is one contiguous 1152-byte run.  Weights are pre-packed (host side)
// as [ntile][C_in/32][tap][128 couts][32 cin] bf16 with the GroupNorm gain folded in and the four 16-byte
// chunks of every 64-byte row XOR-swizzled by ((cout >> 2) & 3), i.e. already in their LDS image.
//
// GroupNorm fold: conv(W, (x-mu)*rstd*g + b) with zero padding applied AFTER the norm equals
//     rstd * conv(W*g, x)  -  rstd*mu * SG[e][o]  +  SA[e][o]
// where SG/SA sum W*g / W*b over the taps that are inside the image for the pixel's edge class e
// (3 row classes x 3 column classes).  The main loop therefore streams raw bf16 activations; the
// per-frame statistics (sum, sum of squares; produced by the previous kernel's epilogue) enter only
// in the epilogue.
//
// Tiling: one workgroup (4 waves) = 16x16 output pixels x 128 output channels of one frame; wave tile
// 128 px x 64 couts = 4x2 MFMA 32x32x16 accumulators.  K loop: for each 32-channel block the 18x18x32
// halo tile is register-staged into LDS once (zero-filled outside the image) and reused by all nine taps;
// the weight tile of one kernel row (3 taps, 24 KB) is DMA'd global->LDS (global_load_lds, no VGPRs, no
// ds_write) into a double buffer one step ahead, so a step costs ONE barrier.  round-1 profile: the first
// version of this kernel was LDS-bound (SQ_LDS_IDX_ACTIVE > MFMA busy, 39 % of it bank conflicts); hence
//   - the weight image is swizzled so ds_read_b128 of B fragments is conflict-free on 64-byte rows,
//   - the 32 rows of an MFMA M-subtile map to pixels so that every hardware 16-lane ds_read_b128 group
//     covers 16 consecutive pixels of ONE image row (80-byte pixel stride -> 16 distinct bank slots),
//   - halo ds_write_b128 are ordered so each 8-lane group hits 8 distinct 16-byte slots.
// Two workgroups per CU (78.6 KB LDS, <= 256 VGPRs) hide each other's barriers and epilogues.
#include "vpt_common.h"
#include "vpt_kernels.h"
#include <stdlib.h>
#include <type_traits>
#ifndef VPT_EPI_NO_CLAMP_RELU
#define VPT_EPI_NO_CLAMP_RELU 0   // A/B builds: 1 = residual modes with the fp32 v_max ReLU of rounds 1-4 (bit-identical outputs)
#endif
#ifndef VPT_EPI_ABLATE
#define VPT_EPI_ABLATE 0   // profiling builds: 1 = no output stores, 2 = no residual loads inside the epilogue, 4 = no residual prefetch in the main loop (registers uninitialised)
#endif
// Profiling switches (VPT_CONV_ABLATE = 1: skip the epilogue, 2: skip the main loop; VPT_CONV_EXTRA_LDS: dynamic LDS to force one
// workgroup per CU) exist ONLY in builds made with -DVPT_CONV_PROFILE (tools/build_variant.sh): the shipped library reads no
// environment variable and its kernel carries no ablation branch -- a stray variable cannot change results.
#ifdef VPT_CONV_PROFILE
#define CONV_ABLATE (a.ablate)
#else
#define CONV_ABLATE 0
#endif

#define A_RS 80
#define A_BYTES (324 * A_RS)            // 25920
#define B_BYTES (3 * 128 * 64)          // 24576 per buffer (unpadded, swizzled)
#define KK_OFF (A_BYTES + 2 * B_BYTES)  // 75072
#define KK_BYTES (9 * 128 * 4)          // 4608
#define BT_OFF (KK_OFF + KK_BYTES + 64)  // 79744: per-channel residual bias of mode 5 (128 floats), after the block statistics' 64 bytes
#define SMEM_BYTES (BT_OFF + 512)       // 80256: two workgroups per CU = 160512 of 163840 bytes
#define PT_RS 272                       // pixel pitch of the 16 x 16 x 128-channel output tile of the pool-fused mode (256 + 16 bytes)
static_assert(256 * PT_RS <= KK_OFF, "the pooled mode's output tile must fit below the epilogue table");

// MFMA M-subtile row i (0..31) -> pixel (row 0/1, col 0..15) of a 2x16 patch.  Rows are swapped for columns
// 4..11 so that each 16-lane ds_read_b128 group {0-3,12-15,20-27} / {4-11,16-19,28-31} stays in one image row.
__device__ __forceinline__ int sub_row(int i) { return ((i >> 4) ^ (i >> 2) ^ (i >> 3)) & 1; }


// The wait in front of each step's barrier is a counted s_waitcnt that retires the weight DMA only and leaves the younger
// halo / residual prefetch loads in flight across the barrier (a plain __syncthreads() drains vmcnt to 0 because an LDS-DMA
// is pending).
// TRACE (tools/conv_trace.py): phase timestamps.  Compile time because s_memrealtime is a scalar-memory operation: one of them
// in flight makes lgkmcnt out of order and every LDS wait of the epilogue degenerates to lgkmcnt(0).
// MODE (compile time, so the epilogue carries no runtime branches): 0 forward, 1 forward + residual,
// 2 dgrad (+ c0 + c1 * xin), 3 dgrad + skip connection, 4 forward + the stack's 3x3 / stride-2 max-pool (CnnDownStack.forward,
// lib/impala_cnn.py:114-117: firstconv -> max_pool2d): the 16 x 16 output tile goes to LDS instead of HBM, its 8 x 8 pooled pixels are
// written -- complete where the 3 x 3 window lies inside the tile, the in-tile part of the maximum on the tile's first pooled row /
// column otherwise -- together with the tile's last row and column (the "seams"); vpt_pool_seam_kernel finishes the seam pixels.
// The pre-pool tensor (2 MB per frame in stack 1: written and read back by vpt_pool_kernel before) never reaches HBM.
// TR (compile time): pixel rows of the workgroup's tile.  16: four waves (2 row groups x 2 channel halves), two workgroups per CU.
// 32: EIGHT waves (4 row groups x 2 channel halves) on a 32 x 16-pixel tile, one workgroup per CU -- the same waves per SIMD, the same
// per-wave program, but the step's 24 KB weight tile is fetched ONCE for 512 pixels instead of once per co-resident workgroup: half
// the weight DMA (288 KB per 256 pixels and K = 1152 before -- the largest stream on a CU's memory path) and
// 11 % less halo (34 x 18 instead of 2 x 18 x 18 pixels).
template <bool TRACE, int MODE, int TR = 16>
__global__ __launch_bounds__(TR * 16, 2) void vpt_conv3x3_kernel(VptConv3x3Args a) {
  static_assert(TR == 16 || TR == 32, "tile rows");
  constexpr int NWAVE = TR / 4, NTHR = 64 * NWAVE;
  constexpr int HPIX = (TR + 2) * 18;                            // halo pixels
  constexpr int NA = (HPIX * 4 + NTHR - 1) / NTHR;               // 16-byte halo chunks per thread and channel block: 6 / 5
  constexpr int NDMA = 24 / NWAVE;                               // 1 KB weight-DMA pieces per wave and step: 6 / 3
  constexpr int A_SZ = HPIX * A_RS, KK_O = A_SZ + 2 * B_BYTES, BT_O = KK_O + KK_BYTES + 64, SMEM_SZ = BT_O + 512;
  constexpr int NKK = (9 * 128 + NTHR - 1) / NTHR;
  static_assert(TR != 16 || (A_SZ == A_BYTES && KK_O == KK_OFF && SMEM_SZ <= SMEM_BYTES), "16-row layout");
  static_assert((MODE != 4 && MODE != 7) || TR == 16, "the pool-fused epilogue is written for 16 x 16 tiles");
  // 5 forward + residual through a per-frame affine:  out = relu(...) + res_scale[f] * res + res_bias[f][channel]  -- the block that
  // follows a stack's GroupNorm `n` reads the pooled tensor itself (already multiplied by n's gain) instead of a normalised copy
  // (lib/impala_cnn.py:118-121 with the `n` pass folded away).
  // 6 dgrad of a block's conv1 that hands the block's conv0 its backward operand directly (round 5):  dy = conv^T + c0 + c1 * xin is the
  // gradient w.r.t. conv0's output y = xin, conv0 has no residual, so its ReLU gate is [xin > 0] and its operand dacc0 = rstd0 * dy * [xin > 0]
  // (rstd0: the statistics of conv0's INPUT, gate_stats) is written instead of dy -- vpt_conv_bwd_prep_kernel's pass over (dy, y) -> dacc
  // shrinks to a reduction over dacc0 alone; gate_u[f] += sum rstd0 * dy * xin (= rstd0 * sum dz v: T1's data term; closed gates add 0).
  constexpr bool GATE = MODE == 6;
  // 7 = mode 4 for the TRAINING forward (round 5): the pooled tensor plus, per pooled value, which window positions hold the maximum -- a 9-bit
  // "differs from the maximum" mask per 16-bit lane (bit 8 - k for scan position k = 3 (dy + 1) + (dx + 1); positions outside the tile or the
  // image: 1), pool_mask [F][Cout/32][H/2][W/2][32] uint16.  The backward routes the pooled gradient to the FIRST zero bit (torch's tie rule)
  // and needs neither the pre-pool tensor nor vpt_pool_kernel (vpt_conv_bwd_prep_pooled_kernel).  Measured cost of the masks: +5 % of the
  // K = 1152 pool-fused launch, +2.5 % of the K = 2304 one (profiles/r05_experiments.md).
  constexpr bool PMASK = MODE == 7;
  constexpr bool PACKED_RELU = MODE == 0;    // forward without residual: ReLU + statistics on the packed 16-bit pairs (see the epilogue)
  // Forward WITH residual (modes 1, 5): the ReLU must precede the residual add in fp32, and gfx950 has no packed fp32 maximum -- but every VALU
  // instruction has a clamp-to-[0, 1] result modifier.  The epilogue's affine step runs SCALED by 2^-40 (rstd and the constant table carry the
  // factor: exact, a power of two commutes with the rounding of the FMA) with the clamp set, which is 2^-40 * ReLU for every value below 2^40, and
  // the residual add that follows is an FMA with 2^40 -- exact product, ONE rounding, bit for bit `ReLU(v) + r`: two packed instructions per value
  // pair where there were a packed FMA, two v_max_f32 and a packed add (192 of the mode's 1121 vector instructions per wave and tile).  Outside
  // [2^-86, 2^40] the result would differ from an unscaled ReLU (denormal / saturated): no GroupNorm-fed convolution output lives there.
  constexpr bool CLAMP_RELU = (MODE == 1 || MODE == 5) && !VPT_EPI_NO_CLAMP_RELU;
  constexpr float RELU_S = CLAMP_RELU ? 0x1p-40f : 1.f, RELU_INV = 0x1p40f;
  constexpr bool BWD = (MODE == 2 || MODE == 3 || MODE == 6), HAS_RES = (MODE == 1 || MODE == 3 || MODE == 5), USE_X = BWD, POOL = (MODE == 4 || MODE == 7), RES_AFF = MODE == 5;
  constexpr bool DEFER_STORES = MODE != 3;   // mode 3 holds skip + xin pieces as well: no registers left for the packed results
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_SZ];
  const int tid = threadIdx.x, lane = tid & 63;
  // profiling (vpt_conv3x3_set_trace): CU id + 100 MHz timestamps of the tile's phases
  int cu_key = -1;
  long long t_trace[3];
  if (TRACE && tid == 0) {
    t_trace[0] = wall_clock64();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID[3:0]
    cu_key = (int)(((xcc & 15u) << 8) | ((hw >> 8) & 0xffu));
  }
#ifdef VPT_CONV_DEPHASE   // profiling builds: the SECOND workgroup of a CU's first generation (wave slot 1 of its SIMD) starts VPT_CONV_DEPHASE x ~4 us late
  if (blockIdx.x < 512 && (__builtin_amdgcn_s_getreg((4 << 11) | 4) & 15u) == 1u)
    for (int i_ = 0; i_ < VPT_CONV_DEPHASE; ++i_) __builtin_amdgcn_s_sleep(127);
#endif
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int tilesX = a.W >> 4, tilesY = a.H / TR;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = L % a.NT; L /= a.NT;
  const int tx = L % tilesX; L /= tilesX;
  const int ty = L % tilesY;
  const int f = L / tilesY;
  const int tx0 = tx * 16, ty0 = ty * TR;
  const int NCB = a.Cin >> 5;
  const int HW = a.H * a.W;

  // ---- halo staging map: chunk q -> (pixel P = 8*(q>>5) + (q&7), part = (q>>3)&3) ----
  int a_loff[NA];            // LDS byte offset, -1: no chunk (beyond the 324 halo pixels)
  unsigned a_gbyte[NA];      // byte offset inside one channel-block plane (clamped to 0 outside the image)
  unsigned a_inside = 0;    // bit m: chunk m lies inside the image (else literal zeros: the conv pads the NORMALISED tensor)
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    const int q = tid + NTHR * m;
    const int P = ((q >> 5) << 3) + (q & 7), part = (q >> 3) & 3;
    a_loff[m] = -1;
    a_gbyte[m] = 0u;
    if (P < HPIX) {
      const int hy = P / 18, hx = P - hy * 18;
      const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
      a_loff[m] = P * A_RS + part * 16;
      if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
        a_gbyte[m] = (unsigned)((y * a.W + x) * 32 + part * 8) * 2u;
        a_inside |= 1u << m;
      }
    }
  }
  const op16_t* xplane = a.x + (size_t)f * NCB * HW * 32;
  // weight DMA: wave w moves pieces (NWAVE*m + w), m = 0..NDMA-1, of the 24 KB step tile; lane = 16-byte chunk
  const op16_t* wbase = a.wpk + (size_t)nt * NCB * 9 * 4096 + (size_t)(w * 64 + lane) * 8;
  unsigned char* bdst = smem + A_SZ + w * 1024;

#define ISSUE_B(step_, buf_)                                                                              \
  do {                                                                                                    \
    const op16_t* wp_ = wbase + (size_t)(step_) * 12288;                                                  \
    _Pragma("unroll") for (int m_ = 0; m_ < NDMA; ++m_)                                                   \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp_ + m_ * (NWAVE * 512)),  \
                                       (__attribute__((address_space(3))) void*)(bdst + (buf_) * B_BYTES + m_ * (NWAVE * 1024)), \
                                       16, 0, 0);                                                         \
  } while (0)

  u32x4 areg[NA];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- prologue: weights of step 0 (DMA), halo of channel block 0, epilogue constant table ----
  ISSUE_B(0, 0);
#pragma unroll
  for (int m = 0; m < NA; ++m) areg[m] = *(const u32x4*)((const char*)xplane + a_gbyte[m]);
  float mean = 0.f, rstd = 1.f, c0f = 0.f, c1f = 0.f;
  const float* esa = a.edge_sa;
  if (!BWD) {
    if (a.kk_frame) {   // the epilogue table of THIS frame was prepared by vpt_nfold_coef_kernel (the input is a pooled tensor whose GroupNorm `n`
      rstd = a.rs_frame[f];                               // is folded into this layer): out = relu(rs * acc + kk_frame[f][e][o])
      esa = a.kk_frame + (size_t)f * 9 * a.CoutPad;
    } else {
      frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
    }
  } else if (a.coef) {
    c0f = a.coef[2 * f];
    c1f = a.coef[2 * f + 1];
  }
  float rgate = 1.f;
  if (GATE) {      // everything the epilogue adds is linear in the scale: fold rstd0 into the coefficients
    float mg;
    frame_mean_rstd(a.gate_stats, f, a.inv_count_gate, mg, rgate);
    c0f *= rgate;
    c1f *= rgate;
  }
  {
    float* kk = (float*)(smem + KK_O);
    float ksa[NKK], ksg[NKK];
#pragma unroll
    for (int k = 0; k < NKK; ++k) {  // 9*128 = 4.5 * 256 entries: all the loads in flight together
      const int idx = tid + NTHR * k;
      const int o = (idx >> 7) * a.CoutPad + nt * 128 + (idx & 127);
      ksa[k] = (idx < 9 * 128 && !BWD) ? esa[o] : 0.f;
      ksg[k] = (idx < 9 * 128 && !BWD) ? a.edge_sg[o] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NKK; ++k) {
      const int idx = tid + NTHR * k;
      if (idx < 9 * 128) kk[idx] = (ksa[k] - rstd * mean * ksg[k]) * RELU_S;
    }
  }
  float res_s = 1.f;
  if (RES_AFF) {
    res_s = a.res_scale[f];
    if (tid < 128) ((float*)(smem + BT_O))[tid] = (nt * 128 + tid < a.Cout) ? a.res_bias[(size_t)f * a.Cout + nt * 128 + tid] : 0.f;
  }
#define WRITE_HALO()                                                                                      \
  _Pragma("unroll") for (int m_ = 0; m_ < NA; ++m_)                                                       \
    if (a_loff[m_] >= 0) *(u32x4*)(smem + a_loff[m_]) = ((a_inside >> m_) & 1u) ? areg[m_] : zero4
  WRITE_HALO();
  __syncthreads();

  f32x16 acc[4][2];
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  // fragment base addresses
  const unsigned char* aL = smem + ((wm * 8 + sub_row(l31)) * 18 + (l31 & 15)) * A_RS + hi * 16;
  const int bsw = (l31 >> 2) & 3;
  const unsigned char* bL0 = smem + A_SZ + (wn * 64 + l31) * 64 + (((0 + hi) ^ bsw) << 4);  // ks = 0
  const unsigned char* bL1 = smem + A_SZ + (wn * 64 + l31) * 64 + (((2 + hi) ^ bsw) << 4);  // ks = 1

  // epilogue addressing (needed early: the residual is requested during the last channel block)
  // Operands are SWAPPED in the MFMA (weights = A rows, pixels = B columns): a lane holds ONE pixel (column l31 of the
  // 2x16-pixel subtile) and, per accumulator, four groups g of 4 consecutive output channels 8g + 4hi .. +3 -- 8 bytes of
  // the pixel's 64-byte channel row per group, the partner lane (l31, 1 - hi) holding the 8 bytes next to them.  One
  // v_permlane32_swap per dword turns the pair (g = 2p, 2p + 1) into 16 CONTIGUOUS bytes per lane (lower half-wave: bytes
  // 32p.., upper: 32p + 16..), and one v_permlane16_swap per dword (lanes l <-> l ^ 16, p = 0 <-> 1) regroups those so that one
  // 16-byte access per lane covers the WHOLE 64-byte channel row of 16 pixels: the residual / xin / output move as complete
  // 128-byte lines, with no LDS staging at all.  (Round 1 staged them through a wave-private LDS tile: four LDS round trips
  // per subtile that queued behind the co-resident workgroup's main-loop LDS traffic.  The first LDS-free version moved 32
  // bytes per pixel and instruction, so every line was touched by two instructions; an ablation without the output stores ran
  // 6-14 % faster, whole-line stores recovered 2-4.4 % of that.)
  const int CB_out = a.Cout >> 5;
  const int cb0 = nt * 4 + wn * 2;                 // 32-channel block of n2 = 0
  const bool nvalid[2] = {(cb0 + 0) < CB_out, (cb0 + 1) < CB_out};
  // Addresses = wave-uniform base (frame, channel block n2) + a 32-bit per-lane byte offset: global accesses with an SGPR
  // base, no 64-bit vector address arithmetic, no address registers kept across the main loop.
  const unsigned gm_b = (unsigned)(2 * a.W) * 64u;   // + m * gm_b: two image rows further down (bytes)
  // instruction j = 0 / 1 of a pair carries the 64-byte rows of the pixels held by lanes (l31 & 15) + 16 j; this lane supplies
  // (receives) bytes 32 (l31 >> 4) + 16 hi .. + 15 of them
  unsigned svoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
    svoff[j] = (unsigned)(((ty0 + wm * 8 + sub_row((l31 & 15) + 16 * j)) * a.W + tx0 + (l31 & 15)) * 32 + 16 * (l31 >> 4) + 8 * hi) * 2u;
  size_t cbase[2];                                  // element offset of channel block n2 of this frame
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) cbase[n2] = (size_t)(f * CB_out + (nvalid[n2] ? cb0 + n2 : 0)) * HW * 32;
#define EPI_LD(ptr_, m_, n2_, p_) (*(const u32x4*)((const char*)((ptr_) + cbase[n2_]) + (svoff[p_] + (unsigned)(m_) * gm_b)))
#ifndef VPT_RES_LATE
#define VPT_RES_LATE 0     // 1: the residual prefetch rides in the tile's last-but-one step (see the last channel block below); 0: one step earlier (rounds 2-4).  Measured: +1.8 % on s0.block alone, nothing on the step (profiles/r04_experiments.md section 15)
#endif
#ifndef VPT_RES_PREFETCH
#define VPT_RES_PREFETCH 1   // 0: no residual request inside the main loop; all four subtiles are requested at the start of the epilogue
#endif
#ifndef VPT_RES_NATIVE
#define VPT_RES_NATIVE 1   // 1: the residual arrives in the accumulators' own layout (8-byte loads: a lane's 4 channels of a group), no lane exchanges;
#endif                     // 0: whole 128-byte lines per 16-byte load + v_permlane16/32_swap (round 2-3; the output stores still go that way)
#if VPT_RES_NATIVE
  // Round 4 (profiles/r04_experiments.md): the residual path cost 16 % of a K = 1152 tile beyond its loads -- 128 of its 256 vector
  // instructions were lane exchanges (43 cycles of latency each, half the issue rate of a plain instruction: tools/ubench/permlane.hip) in
  // front of every add.  An 8-byte load per (subtile, channel block, group) puts the lane's own 4 channels where the add needs them.
  u32x2 rq[4][2][4];                                // residual [subtile m][n2][group g]
  const unsigned nvoff = (unsigned)(((ty0 + wm * 8 + sub_row(l31)) * a.W + tx0 + (l31 & 15)) * 32 + 4 * hi) * 2u;
#define EPI_LDN(ptr_, m_, n2_, g_) (*(const u32x2*)((const char*)((ptr_) + cbase[n2_]) + (nvoff + (unsigned)(m_) * gm_b + 16u * (unsigned)(g_))))
#define LOAD_RES(m_)                                                                                      \
  _Pragma("unroll") for (int n2_ = 0; n2_ < 2; ++n2_)                                                     \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) rq[m_][n2_][g_] = EPI_LDN(a.res, m_, n2_, g_)
#else
  u32x4 rq[4][2][2];                                // residual [subtile m][n2][instruction j of the pair]
#define LOAD_RES(m_)                                                                                      \
  _Pragma("unroll") for (int n2_ = 0; n2_ < 2; ++n2_)                                                     \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) rq[m_][n2_][p_] = EPI_LD(a.res, m_, n2_, p_)
#endif

  // ---- main loop ----------------------------------------------------------------------------------------------------
  // One K step = one kernel row (3 taps) of one 32-channel block = 6 groups (tap dx, 16-channel half ks) of 8 MFMAs per
  // wave.  Round-1 profile: 55 % of the wave cycles were SQ_WAIT_INST_ANY -- every group read its six fragments into the
  // registers the previous group's MFMAs had just released and then waited lgkmcnt(0), so the matrix pipe drained for one
  // LDS round trip per 8 MFMAs (one workgroup per CU alone reached 85 % of two).  Now the schedule is explicit, one
  // instruction pair at a time, pinned with sched_barrier (left alone, the scheduler sinks every fragment read next to
  // its first use to save registers):
  //   - two fragment register sets; while group g's MFMAs issue, the fragments of group g+1 are requested, one
  //     pair of ds_read_b128 behind each of the first three MFMAs, in the order the next group consumes them;
  //   - the two remaining slots of each group carry the step's global traffic: the weight DMA of the next step
  //     (6 x global_load_lds, groups 0-2) and the halo of the next channel block / the residual (groups 3-4), so no
  //     MFMA ever queues behind a burst of memory instructions;
  //   - the step's barrier sits in front of the LAST group's MFMAs: every wave then holds its last fragments of the
  //     step in registers, the next step's weights (requested >= 24 MFMAs earlier) have landed, so the next step's
  //     first fragments are requested behind the barrier and arrive under group 5's MFMAs.
  op16x8 fa[2][4], fb[2][2];
  const op16_t* resp = a.res;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MM(set_, m_, n_) acc[m_][n_] = VPT_MFMA_32X32X16(fb[set_][n_], fa[set_][m_], acc[m_][n_], 0, 0, 0)
#define MMZ(set_, m_, n_) acc[m_][n_] = VPT_MFMA_32X32X16(fb[set_][n_], fa[set_][m_], zero16, 0, 0, 0)   /* C = 0: no zero-initialised accumulators */
#define FA_LD(set_, dy_, g_, m_) \
  fa[set_][m_] = *(const op16x8*)(aL + ((dy_) * 18 + ((g_) >> 1)) * A_RS + (m_) * (2 * 18 * A_RS) + ((g_) & 1) * 32)
#define FB_LD(set_, g_, n_, boff_) \
  fb[set_][n_] = *(const op16x8*)((((g_) & 1) ? bL1 : bL0) + (boff_) + ((g_) >> 1) * (128 * 64) + (n_) * (32 * 64))
  // MFMAs of register set `set_`; fragments of group (ndy_, ng_) go to the other set; X0 / X1: the two free slots
#define GROUP_(MM_, set_, ndy_, ng_, nboff_, X0, X1)                                                      \
  do {                                                                                                    \
    MM_(set_, 0, 0); FB_LD(1 - (set_), ng_, 0, nboff_); FA_LD(1 - (set_), ndy_, ng_, 0); SB();            \
    MM_(set_, 0, 1); FB_LD(1 - (set_), ng_, 1, nboff_); FA_LD(1 - (set_), ndy_, ng_, 1); SB();            \
    MM_(set_, 1, 0); FA_LD(1 - (set_), ndy_, ng_, 2); FA_LD(1 - (set_), ndy_, ng_, 3); SB();              \
    MM_(set_, 1, 1); SB();                                                                                \
    MM_(set_, 2, 0); X0; SB();                                                                            \
    MM_(set_, 2, 1); SB();                                                                                \
    MM_(set_, 3, 0); X1; SB();                                                                            \
    MM_(set_, 3, 1); SB();                                                                                \
  } while (0)
#define GROUP(set_, ndy_, ng_, nboff_, X0, X1) GROUP_(MM, set_, ndy_, ng_, nboff_, X0, X1)
#define GROUP_Z(set_, ndy_, ng_, nboff_, X0, X1) GROUP_(MMZ, set_, ndy_, ng_, nboff_, X0, X1)   /* the tile's first 8 MFMAs */
#define GROUP_TAIL(set_)                                                                                  \
  do {                                                                                                    \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) { MM(set_, m_, 0); MM(set_, m_, 1); }                \
  } while (0)
#ifndef VPT_CONV_HALO_ABLATE
#define VPT_CONV_HALO_ABLATE 0  // profiling builds (wrong results, timing only): 1 = halo written without the zero-fill selects, 2 = no halo loads / writes after block 0
#endif
#ifndef VPT_CONV_DMA_PIECES
#define VPT_CONV_DMA_PIECES 6   // profiling builds: fewer weight-DMA pieces per wave and step (the results are then wrong; timing only)
#endif
#define GLDS(m_)                                                                                          \
  do { if ((m_) < VPT_CONV_DMA_PIECES && (m_) < NDMA)                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp_ + (m_) * (NWAVE * 512)),    \
                                   (__attribute__((address_space(3))) void*)(bd_ + (m_) * (NWAVE * 1024)), 16, 0, 0); } while (0)
#define XA(m_) do { if (VPT_CONV_HALO_ABLATE != 2 && (m_) < NA) areg[(m_) < NA ? (m_) : 0] = *(const u32x4*)((const char*)xplane + (cbo_ + a_gbyte[m_])); } while (0)
#if VPT_RES_NATIVE
#define XR(m_, n2_, p_) do { rq[m_][n2_][2 * (p_)] = EPI_LDN(resp, m_, n2_, 2 * (p_)); rq[m_][n2_][2 * (p_) + 1] = EPI_LDN(resp, m_, n2_, 2 * (p_) + 1); } while (0)
#else
#define XR(m_, n2_, p_) rq[m_][n2_][p_] = EPI_LD(resp, m_, n2_, p_)
#endif
#define NOP_() ((void)0)
#define WAIT_BARRIER(n_late_)                                                                             \
  do {                                                                                                    \
    asm volatile("s_waitcnt vmcnt(" #n_late_ ") lgkmcnt(0)" ::: "memory");                                \
    __builtin_amdgcn_s_barrier();                                                                         \
    asm volatile("" ::: "memory");                                                                        \
    SB();                                                                                                 \
  } while (0)
  // Memory ops are issued in a FIXED order and count per wave -- weight DMA (6), then either the halo of the next block
  // (6, PRE_A) or, in the modes with a residual, subtiles 0 / 1 of it (8, PRE_R) -- so the counted wait in front of the barrier
  // is exact: it retires the DMA and leaves the younger loads in flight across the barrier.  Every counted load must be LIVE
  // (a load whose result is unused is deleted by the compiler and the count would then release the barrier early -- this
  // bit once: the residual prefetch of a residual-free instantiation), hence the compile-time HAS_RES in the step.
#define CONV_STEP(cb_, dy_, NEXT_DY, PRE_A, WR_A, PRE_R, LAST, FIRST)                                            \
  do {                                                                                                    \
    const int s_ = (cb_) * 3 + (dy_);                                                                     \
    const int boff_ = (s_ & 1) * B_BYTES, noff_ = B_BYTES - boff_;                                        \
    const op16_t* wp_ = wbase + (size_t)(s_ + 1) * 12288;                                                 \
    unsigned char* bd_ = bdst + noff_;                                                                    \
    const unsigned cbo_ = (unsigned)((cb_) + 1) * (unsigned)HW * 64u; /* bytes; uniform */                \
    if (FIRST) {                                                                                          \
      GROUP_Z(0, dy_, 1, boff_, GLDS(0), GLDS(1));                                                        \
      GROUP(1, dy_, 2, boff_, GLDS(2), GLDS(3));                                                          \
      GROUP(0, dy_, 3, boff_, GLDS(4), GLDS(5));                                                          \
    } else if (LAST) {                                                                                    \
      GROUP(0, dy_, 1, boff_, NOP_(), NOP_());                                                            \
      GROUP(1, dy_, 2, boff_, NOP_(), NOP_());                                                            \
      GROUP(0, dy_, 3, boff_, NOP_(), NOP_());                                                            \
    } else {                                                                                              \
      GROUP(0, dy_, 1, boff_, GLDS(0), GLDS(1));                                                          \
      GROUP(1, dy_, 2, boff_, GLDS(2), GLDS(3));                                                          \
      GROUP(0, dy_, 3, boff_, GLDS(4), GLDS(5));                                                          \
    }                                                                                                     \
    if (PRE_A) {                                                                                          \
      GROUP(1, dy_, 4, boff_, do { XA(0); XA(1); } while (0), do { XA(2); } while (0));                   \
      GROUP(0, dy_, 5, boff_, do { XA(3); XA(4); } while (0), do { XA(5); } while (0));                   \
      if (VPT_CONV_HALO_ABLATE == 2) WAIT_BARRIER(0); else if (NA == 6) WAIT_BARRIER(6); else WAIT_BARRIER(5);                               \
    } else if ((PRE_R) && HAS_RES) { /* compile-time: without a residual the loads would be dead code and the count wrong */ \
      GROUP(1, dy_, 4, boff_, do { XR(0, 0, 0); XR(0, 0, 1); } while (0), do { XR(0, 1, 0); XR(0, 1, 1); } while (0)); \
      GROUP(0, dy_, 5, boff_, do { XR(1, 0, 0); XR(1, 0, 1); } while (0), do { XR(1, 1, 0); XR(1, 1, 1); } while (0)); \
      if (VPT_RES_NATIVE) WAIT_BARRIER(16); else WAIT_BARRIER(8);   /* (two 8-byte loads per XR in the native layout) */ \
    } else {                                                                                              \
      GROUP(1, dy_, 4, boff_, NOP_(), NOP_());                                                            \
      GROUP(0, dy_, 5, boff_, NOP_(), NOP_());                                                            \
      /* the tile's last step issues no DMA: the residual pieces requested in the step before stay in flight across its barrier */ \
      if ((LAST) && HAS_RES && VPT_RES_LATE && VPT_RES_PREFETCH && !(VPT_EPI_ABLATE & 4)) { if (VPT_RES_NATIVE) WAIT_BARRIER(16); else WAIT_BARRIER(8); }  \
      else WAIT_BARRIER(0);                                                                               \
    }                                                                                                     \
    if (WR_A) {   /* the halo of the next channel block replaces the current one: second barrier before its first read */ \
      _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                                  \
        if (VPT_CONV_HALO_ABLATE == 0) {                                                                  \
          if (a_loff[m_] >= 0) *(u32x4*)(smem + a_loff[m_]) = ((a_inside >> m_) & 1u) ? areg[m_] : zero4; \
          if (m_ + 4 < NA && a_loff[m_ + 4 < NA ? m_ + 4 : 0] >= 0) *(u32x4*)(smem + a_loff[m_ + 4 < NA ? m_ + 4 : 0]) = ((a_inside >> (m_ + 4)) & 1u) ? areg[m_ + 4 < NA ? m_ + 4 : 0] : zero4; \
        } else if (VPT_CONV_HALO_ABLATE == 1) {                                                           \
          if (a_loff[m_] >= 0) *(u32x4*)(smem + a_loff[m_]) = areg[m_];                                   \
          if (m_ + 4 < NA && a_loff[m_ + 4 < NA ? m_ + 4 : 0] >= 0) *(u32x4*)(smem + a_loff[m_ + 4 < NA ? m_ + 4 : 0]) = areg[m_ + 4 < NA ? m_ + 4 : 0];         \
        }                                                                                                 \
        SB(); MM(1, m_, 0); SB();                                                                         \
      }                                                                                                   \
      WAIT_BARRIER(0);                                                                                    \
      MM(1, 0, 1); FB_LD(0, 0, 0, noff_); SB();                                                           \
      MM(1, 1, 1); FA_LD(0, 0, 0, 0); SB();                                                               \
      MM(1, 2, 1); FB_LD(0, 0, 1, noff_); FA_LD(0, 0, 0, 1); SB();                                        \
      MM(1, 3, 1); FA_LD(0, 0, 0, 2); FA_LD(0, 0, 0, 3); SB();                                            \
    } else if (LAST) {                                                                                    \
      GROUP_TAIL(1);                                                                                      \
    } else {                                                                                              \
      GROUP(1, NEXT_DY, 0, noff_, NOP_(), NOP_());                                                        \
    }                                                                                                     \
  } while (0)

  if (TRACE && tid == 0) t_trace[1] = wall_clock64();
  if (CONV_ABLATE != 2) {
    FB_LD(0, 0, 0, 0); FA_LD(0, 0, 0, 0); FB_LD(0, 0, 1, 0); FA_LD(0, 0, 0, 1); FA_LD(0, 0, 0, 2); FA_LD(0, 0, 0, 3);
    SB();
    if (NCB > 1) {   // first channel block peeled: its first eight MFMAs take C = 0 instead of 128 zeroed registers
      CONV_STEP(0, 0, 1, true, false, false, false, true);
      CONV_STEP(0, 1, 2, false, false, false, false, false);
      CONV_STEP(0, 2, 0, false, true, false, false, false);
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    }
    for (int cb = 1; cb + 1 < NCB; ++cb) {
      CONV_STEP(cb, 0, 1, true, false, false, false, false);
      CONV_STEP(cb, 1, 2, false, false, false, false, false);
      CONV_STEP(cb, 2, 0, false, true, false, false, false);
    }
    // last channel block: no further halo -> request the residual of the first two subtiles instead.  A step's barrier waits for its weight DMA with
    // a counted vmcnt, which retires everything issued BEFORE that DMA too: a residual requested in kernel row 0 is forced home by row 1's barrier (56
    // MFMAs later, workgroup-wide); requested in row 1 behind that step's DMA (VPT_RES_LATE) nothing waits for it before the epilogue, because the last
    // step issues no DMA.  Built and measured in round 4: +1.8 % on the K = 1152 layer's micro-benchmark, 0.1 % on the forward step -- the default stays.
    CONV_STEP(NCB - 1, 0, 1, false, false, !VPT_RES_LATE && !(VPT_EPI_ABLATE & 4) && VPT_RES_PREFETCH, false, false);
    CONV_STEP(NCB - 1, 1, 2, false, false, VPT_RES_LATE != 0 && !(VPT_EPI_ABLATE & 4) && VPT_RES_PREFETCH, false, false);
    CONV_STEP(NCB - 1, 2, 0, false, false, false, true, false);
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    __syncthreads();
  }
#undef CONV_STEP
#undef WAIT_BARRIER
#undef GROUP
#undef GROUP_TAIL
#undef GLDS
#undef XA
#undef XR
#undef MM
#undef MMZ
#undef GROUP_
#undef GROUP_Z
#undef FA_LD
#undef FB_LD

  // ---------------- epilogue ----------------
  if (TRACE && tid == 0) t_trace[2] = wall_clock64();
  if (CONV_ABLATE == 1) {  // profiling: keep the accumulators live, skip the epilogue
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[m][n][r];
    if (t == 12345.678f) a.y[0] = (vpt_op16)t;
    return;
  }
  if (HAS_RES && (CONV_ABLATE == 2 || !VPT_RES_PREFETCH)) { LOAD_RES(0); LOAD_RES(1); }
  SB();

  f32x2 s_sum2 = {0.f, 0.f}, s_sq2 = {0.f, 0.f};   // packed fp32 (v_pk_add_f32 / v_pk_fma_f32): two values per VALU issue
  // The body is instantiated for NV = 2 and NV = 1 valid 32-channel blocks of this wave (Cout / 32 odd) and selected by one
  // uniform branch: with no control flow inside, the waits on the prefetched table / residual pieces stay counted.
  auto epilogue = [&](auto nv_) __attribute__((always_inline)) {
  constexpr int NV = decltype(nv_)::value;
  int eoff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int y = ty0 + wm * 8 + 2 * m + sub_row(l31);
    const int x = tx0 + (l31 & 15);
    const int ey = (y == 0) ? 0 : ((y == a.H - 1) ? 2 : 1);
    const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
    eoff[m] = (ey * 3 + ex) * 128 + wn * 64 + 4 * hi;
  }
  const float* kk = (const float*)(smem + KK_O);
  constexpr bool use_x = USE_X;
  u32x4 xq[2][2][2];          // dgrad: the forward layer's input, [parity of m][n2][pair], one subtile ahead
  u32x4 outv[4][2][2];        // packed results: ALL stores are issued after the last load has been consumed.  gfx950 has one
                              // counter (vmcnt) for loads and stores, which may retire out of order relative to each other, so a
                              // wait for a load issued among stores degenerates to vmcnt(0) = "every store has reached L2" --
                              // the round-2 trace showed the first subtile waiting 4-14 us that way.
  // Forward: the constant table of the GroupNorm fold, one subtile AHEAD.  Read at the point of use (two ds_read_b128, wait,
  // eight FMAs) every pair paid a full LDS round trip behind the co-resident workgroup's fragment reads -- 32 exposed round
  // trips per tile, which is why an epilogue beside a main loop took twice as long as one beside another epilogue and the two
  // workgroups of a CU re-locked their phases within one tile even when started half a period apart.
  f32x4 kq[2][4];             // [parity of chunk c = 2 m + n2][channel group g], one chunk (half a subtile, ~150 VALU) ahead
#define LOAD_KK(c_)                                                                                       \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) kq[(c_) & 1][g_] = *(const f32x4*)(kk + eoff[(c_) >> 1] + ((c_) & 1) * 32 + 8 * g_)
#define LOAD_XIN(m_)                                                                                      \
  _Pragma("unroll") for (int n2_ = 0; n2_ < 2; ++n2_)                                                     \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) xq[(m_) & 1][n2_][p_] = EPI_LD(a.xin, m_, n2_, p_)
  if (use_x) LOAD_XIN(0);
  if (!BWD) LOAD_KK(0);
  SB();

  // 16-byte piece {first half x, y | second half z, w} of a lane pair -> this lane's own 8 bytes of group 2p (e) and 2p + 1 (o)
#define UNSWAP(v_, e_, o_)                                                                                \
  do {                                                                                                    \
    const auto s0_ = __builtin_amdgcn_permlane32_swap((v_).x, (v_).z, false, false);                      \
    const auto s1_ = __builtin_amdgcn_permlane32_swap((v_).y, (v_).w, false, false);                      \
    (e_).x = s0_[0]; (o_).x = s0_[1]; (e_).y = s1_[0]; (o_).y = s1_[1];                                   \
  } while (0)
  const f32x2 zero2 = {0.f, 0.f};
  const f32x2 rstd2 = {rstd * RELU_S, rstd * RELU_S}, c0f2 = {c0f, c0f}, c1f2 = {c1f, c1f}, ress2 = {res_s, res_s}, rgate2 = {rgate, rgate};
  const f32x2 relu_inv2 = {RELU_INV, RELU_INV};
  // mode 5: the residual's per-channel bias of this lane's channels.  Read per chunk (4 x ds_read_b128 right behind the chunk's table prefetch, ~20
  // VALU instructions before their first use) instead of held for the whole epilogue: 16 registers instead of 32 -- round 4's version spilled 2 VGPRs.
  f32x4 bq[4];
  const float* btab = (const float*)(smem + BT_O) + wn * 64 + 4 * hi;

#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (HAS_RES && m < 2 && !(VPT_EPI_ABLATE & 2)) LOAD_RES(m + 2);   // rolling prefetch, two subtiles ahead (0 / 1 were requested during the last channel block)
    if (use_x && m < 3) LOAD_XIN(m + 1);   // next subtile's xin: in flight while this one is processed
    SB();
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      if (!BWD && 2 * m + n2 < 7) { LOAD_KK(2 * m + n2 + 1); SB(); }
      if (n2 >= NV) continue;
      if (RES_AFF) {
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) bq[g_] = *(const f32x4*)(btab + n2 * 32 + 8 * g_);
        SB();
      }
      u32x4 ovp[2], xp[2];
#if !VPT_RES_NATIVE
      u32x4 rp[2];
#endif
      // residual / xin arrive as whole pixel rows (instruction j: the pixels of lanes (l31 & 15) + 16 j); the same exchange as for
      // the stores, run backwards, gives every lane the two 16-byte pieces (p = 0, 1) of its own pixel
#define ROWS_TO_PIECES(src_, dst_)                                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
    const auto sw_ = __builtin_amdgcn_permlane16_swap((src_)[0][j_], (src_)[1][j_], false, false);        \
    (dst_)[0][j_] = sw_[0]; (dst_)[1][j_] = sw_[1];                                                       \
  }
#if !VPT_RES_NATIVE
      if (HAS_RES) ROWS_TO_PIECES(rq[(VPT_EPI_ABLATE & 2) ? (m & 1) : m][n2], rp);
#endif
      if (use_x) ROWS_TO_PIECES(xq[m & 1][n2], xp);
#undef ROWS_TO_PIECES
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        u32x2 r2[2] = {{0u, 0u}, {0u, 0u}}, x2[2] = {{0u, 0u}, {0u, 0u}}, pk[2];
#if VPT_RES_NATIVE
        if (HAS_RES) { r2[0] = rq[(VPT_EPI_ABLATE & 2) ? (m & 1) : m][n2][2 * p]; r2[1] = rq[(VPT_EPI_ABLATE & 2) ? (m & 1) : m][n2][2 * p + 1]; }
#else
        if (HAS_RES) UNSWAP(rp[p], r2[0], r2[1]);
#endif
        if (use_x) UNSWAP(xp[p], x2[0], x2[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int g = 2 * p + q;
          f32x2 v01 = {acc[m][n2][4 * g + 0], acc[m][n2][4 * g + 1]}, v23 = {acc[m][n2][4 * g + 2], acc[m][n2][4 * g + 3]};
          if (!BWD) {
            const f32x4 k4 = kq[n2][g];
            const f32x2 k01 = {k4.x, k4.y}, k23 = {k4.z, k4.w};
            if (PACKED_RELU) {   // the ReLU follows the 16-bit pack (below): there is no packed fp32 maximum on gfx950, two v_max_f32 per pair were a quarter of this mode's epilogue
              v01 = rstd2 * v01 + k01;
              v23 = rstd2 * v23 + k23;
            } else if (CLAMP_RELU) {   // 2^-40 * ReLU(rstd v + k) through the clamp modifier (see CLAMP_RELU above)
              v01 = pk_fma_clamp01(rstd2, v01, k01);
              v23 = pk_fma_clamp01(rstd2, v23, k23);
            } else {
              v01 = __builtin_elementwise_max(rstd2 * v01 + k01, zero2);
              v23 = __builtin_elementwise_max(rstd2 * v23 + k23, zero2);
            }
          } else if (use_x) {  // dgrad: + d(mu, rstd)/dx terms of the GroupNorm statistics
            const f32x2 x01 = {op16_lo_to_f32(x2[q].x), op16_hi_to_f32(x2[q].x)}, x23 = {op16_lo_to_f32(x2[q].y), op16_hi_to_f32(x2[q].y)};
            if (GATE) {
              v01 = rgate2 * v01 + (c1f2 * x01 + c0f2);     // rstd0 * dy  (c0, c1 carry rstd0 already)
              v23 = rgate2 * v23 + (c1f2 * x23 + c0f2);
              s_sum2 = v01 * x01 + s_sum2;                  // sum rstd0 dy xin: where the gate is closed xin = 0 contributes nothing
              s_sq2 = v23 * x23 + s_sq2;
              v01.x = x01.x > 0.f ? v01.x : 0.f; v01.y = x01.y > 0.f ? v01.y : 0.f;
              v23.x = x23.x > 0.f ? v23.x : 0.f; v23.y = x23.y > 0.f ? v23.y : 0.f;
            } else {
              v01 += c1f2 * x01 + c0f2;
              v23 += c1f2 * x23 + c0f2;
            }
          }
          if (HAS_RES) {
            const f32x2 r01 = {op16_lo_to_f32(r2[q].x), op16_hi_to_f32(r2[q].x)}, r23 = {op16_lo_to_f32(r2[q].y), op16_hi_to_f32(r2[q].y)};
            if (RES_AFF) {
              const f32x4 b4 = bq[g];
              const f32x2 b01 = {b4.x, b4.y}, b23 = {b4.z, b4.w};
              if (CLAMP_RELU) {
                v01 = ress2 * r01 + __builtin_elementwise_fma(v01, relu_inv2, b01);
                v23 = ress2 * r23 + __builtin_elementwise_fma(v23, relu_inv2, b23);
              } else {
                v01 = ress2 * r01 + (v01 + b01);
                v23 = ress2 * r23 + (v23 + b23);
              }
            } else if (CLAMP_RELU) {
              v01 = __builtin_elementwise_fma(v01, relu_inv2, r01);
              v23 = __builtin_elementwise_fma(v23, relu_inv2, r23);
            } else {
              v01 += r01;
              v23 += r23;
            }
          }
          pk[q].x = pack_op16x2(v01.x, v01.y);
          pk[q].y = pack_op16x2(v23.x, v23.y);
          if (!BWD && !PACKED_RELU) {   // frame statistics of the output (the next layer's GroupNorm), from the stored pairs; dgrad has no consumer for them
            s_sum2.x = dot2_op16(pk[q].x, OP16_ONE2, s_sum2.x);
            s_sum2.y = dot2_op16(pk[q].y, OP16_ONE2, s_sum2.y);
            s_sq2.x = dot2_op16(pk[q].x, pk[q].x, s_sq2.x);
            s_sq2.y = dot2_op16(pk[q].y, pk[q].y, s_sq2.y);
          }
          if (PACKED_RELU) {
            // Round 5: ReLU as ONE packed signed-16-bit maximum per pair on the rounded bit patterns (positive 16-bit floats
            // order like integers, negative ones -- and -0 -- are negative integers; rounding is monotone, so max(round(v), 0) == round(max(v, 0)):
            // bit-identical outputs), and the frame statistics from the packed pair by v_dot2 (products of two 16-bit operands are exact in fp32):
            // 5 instructions per value pair instead of 6, and the statistics are those of the STORED tensor -- what the next layer's GroupNorm
            // normalises and what the reference's nn.GroupNorm sees.  Two independent accumulator chains per moment.
            pk[q].x = relu_op16x2(pk[q].x);
            pk[q].y = relu_op16x2(pk[q].y);
            s_sum2.x = dot2_op16(pk[q].x, OP16_ONE2, s_sum2.x);
            s_sum2.y = dot2_op16(pk[q].y, OP16_ONE2, s_sum2.y);
            s_sq2.x = dot2_op16(pk[q].x, pk[q].x, s_sq2.x);
            s_sq2.y = dot2_op16(pk[q].y, pk[q].y, s_sq2.y);
          }
        }
        // back to 16 contiguous bytes per lane (the swap is an involution) and out
        const auto o0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto o1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        ovp[p] = u32x4{o0[0], o1[0], o0[1], o1[1]};
      }
      // One more exchange, between lanes l and l ^ 16 (v_permlane16_swap), so that each store instruction writes all 64 bytes of
      // 16 pixels -- whole 128-byte lines -- instead of one 32-byte half of 32 pixels' rows (every line was then written by two
      // instructions: twice the write requests on the CU's path to L2).
      {
        u32x4 oa, ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane16_swap(ovp[0][j], ovp[1][j], false, false);
          oa[j] = sw[0]; ob[j] = sw[1];
        }
        if (DEFER_STORES) { outv[m][n2][0] = oa; outv[m][n2][1] = ob; }
        else {
          *(u32x4*)((char*)(a.y + cbase[n2]) + (svoff[0] + (unsigned)m * gm_b)) = oa;
          *(u32x4*)((char*)(a.y + cbase[n2]) + (svoff[1] + (unsigned)m * gm_b)) = ob;
        }
      }
    }
  }
#undef LOAD_KK
#undef UNSWAP
#undef LOAD_XIN
  if (DEFER_STORES)
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n2 = 0; n2 < NV; ++n2) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (!(VPT_EPI_ABLATE & 1) || s_sum2.x == 12345.678f) *(u32x4*)((char*)(a.y + cbase[n2]) + (svoff[p] + (unsigned)m * gm_b)) = outv[m][n2][p];
    }
  };
  if constexpr (POOL) {
    // The four pointers only this epilogue uses are read from the kernarg segment HERE, through a pointer the compiler cannot see behind: taken from
    // `a` they are loaded with the rest of the arguments at kernel entry and sit in eight scalar registers through the whole main loop -- in the
    // arg-max mode (7) that was the eight registers too many (8 SGPRs spilled to VGPR lanes around the loop).
    const __attribute__((address_space(4))) VptConv3x3Args* late = (const __attribute__((address_space(4))) VptConv3x3Args*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(late));
    vpt_op16* const seam_r_p = late->seam_r;
    vpt_op16* const seam_c_p = late->seam_c;
    vpt_op16* const pool_mask_p = late->pool_mask;
    double* const chs_out_p = late->chs_out;
    // ---- phase 1: GroupNorm fold + ReLU, rounded to 16 bits, into the LDS tile [16 x 16 pixels][128 channels] (pixel pitch PT_RS: the
    // 16 extra bytes spread a column of pixels over the banks).  The tile reuses the halo / weight buffers: every wave must be past
    // its last fragment read first.  A lane holds 4 consecutive channels of one pixel per accumulator group: one ds_write_b64 each.
    __syncthreads();
    {
      const float* kk = (const float*)(smem + KK_O);
      const f32x2 zero2 = {0.f, 0.f};
      const f32x2 rstd2 = {rstd, rstd};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int py = wm * 8 + 2 * m + sub_row(l31), px = l31 & 15;
        const int y = ty0 + py, x = tx0 + px;
        const int ey = (y == 0) ? 0 : ((y == a.H - 1) ? 2 : 1);
        const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
        const float* ke = kk + (ey * 3 + ex) * 128 + wn * 64 + 4 * hi;
        unsigned char* dst = smem + (py * 16 + px) * PT_RS + (wn * 64 + 4 * hi) * 2;
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          if (!nvalid[n2]) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 k4 = *(const f32x4*)(ke + n2 * 32 + 8 * g);
            const f32x2 k01 = {k4.x, k4.y}, k23 = {k4.z, k4.w};
            f32x2 v01 = {acc[m][n2][4 * g + 0], acc[m][n2][4 * g + 1]}, v23 = {acc[m][n2][4 * g + 2], acc[m][n2][4 * g + 3]};
            if (PMASK) {      // (the masks compare post-ReLU values: a window of negatives must read as a window of zeros)
              v01 = __builtin_elementwise_max(rstd2 * v01 + k01, zero2);
              v23 = __builtin_elementwise_max(rstd2 * v23 + k23, zero2);
            } else {          // inference: the pool's packed maximum starts from 0 and the seam kernel's from a pooled value >= 0 -- that IS the ReLU
              v01 = rstd2 * v01 + k01;
              v23 = rstd2 * v23 + k23;
            }
            const u32x2 pk = {pack_op16x2(v01.x, v01.y), pack_op16x2(v23.x, v23.y)};
            *(u32x2*)(dst + (n2 * 32 + 8 * g) * 2) = pk;
          }
        }
      }
    }
    __syncthreads();
    // ---- phase 2: pooled pixel (j, i) of the tile = max over conv rows 2j-1 .. 2j+1, columns 2i-1 .. 2i+1 that lie INSIDE the tile (j = 0
    // / i = 0 miss the row / column above / left of the tile: image border -> the pool's padding, nothing is missing; otherwise the seam
    // kernel adds it).  Packed signed 16-bit max from 0 on the raw bit patterns (post-ReLU values are >= +0; a -0.0 stays below 0).
    // item = (pooled pixel, channel octet): the 16 lanes of a ds_read_b128 group read the 256 contiguous bytes of one pixel.
    typedef short i16x8 __attribute__((ext_vector_type(8)));
    const int PH = a.H >> 1, PW = a.W >> 1;
    const bool top_open = ty0 > 0, left_open = tx0 > 0;      // the tile's first pooled row / column is incomplete
    float p_sum = 0.f, p_sq = 0.f;
    float c1[8], c2[8];      // per-channel sums of the STORED (scaled, rounded) complete pixels: a thread's four items share the octet tid & 15
#pragma unroll
    for (int k = 0; k < 8; ++k) { c1[k] = 0.f; c2[k] = 0.f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int item = tid + 256 * it;
      const int oct = item & 15, pi = (item >> 4) & 7, pj = item >> 7;
      const int cg = nt * 128 + oct * 8;
      const unsigned char* src = smem + oct * 16;
      i16x8 mx = {0, 0, 0, 0, 0, 0, 0, 0};
      i16x8 vv[PMASK ? 9 : 1];
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int r = 2 * pj + dy;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int c = 2 * pi + dx;
          const i16x8 v = *(const i16x8*)(src + (max(r, 0) * 16 + max(c, 0)) * PT_RS);     // (clamped: the duplicate does not change a maximum)
          mx = __builtin_elementwise_max(mx, v);
          if (PMASK) vv[PMASK ? (dy + 1) * 3 + dx + 1 : 0] = v;
        }
      }
      if constexpr (PMASK) {
        // which positions hold the maximum: per position and channel pair xor, packed min with 1 (0: equal), m = 2 m + b -- three packed
        // instructions (inline asm: left to the compiler the 16-bit lanes are scalarised into v_cmp_ne_u16 + v_cndmask pairs, 930 instructions
        // per thread instead of 430).  Positions outside the tile (clamped reads above: duplicates) or the image count as "differs"; the seam
        // kernel fills in the bits of the neighbouring tiles' row / column for the pixels it finishes.
        const u32x4 mxu = __builtin_bit_cast(u32x4, mx);
        u32x4 mk = {0u, 0u, 0u, 0u};
        const uint32_t one2 = 0x00010001u, two2 = 0x00020002u;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const u32x4 vk = __builtin_bit_cast(u32x4, vv[k]);
          const bool inside = (2 * pj + k / 3 - 1 >= 0) && (2 * pi + k % 3 - 1 >= 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint32_t t = inside ? (vk[j] ^ mxu[j]) : 0xffffffffu, b, m2;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(b) : "v"(t), "s"(one2));
            asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(m2) : "v"(mk[j]), "s"(two2), "v"(b));
            mk[j] = m2;
          }
        }
        if (cg < a.Cout) {
          const size_t moff = ((size_t)(f * CB_out + (cg >> 5)) * PH * PW + (size_t)(((ty0 >> 1) + pj) * PW + (tx0 >> 1) + pi)) * 32 + (cg & 31);
          *(u32x4*)(pool_mask_p + moff) = mk;
        }
      }
      if (cg < a.Cout) {
        u32x4 mv = __builtin_bit_cast(u32x4, mx);
        const size_t off = ((size_t)(f * CB_out + (cg >> 5)) * PH * PW + (size_t)(((ty0 >> 1) + pj) * PW + (tx0 >> 1) + pi)) * 32 + (cg & 31);
        const bool complete = (pj > 0 || !top_open) && (pi > 0 || !left_open);
        if (complete) {      // the seam kernel accounts for the others once they are final
#pragma unroll
          for (int k = 0; k < 4; ++k) {     // packed pairs: one v_dot2 per two values and moment (products of 16-bit operands are exact in fp32)
            p_sum = dot2_op16(mv[k], OP16_ONE2, p_sum);
            p_sq = dot2_op16(mv[k], mv[k], p_sq);
          }
          if (a.out_gain) {  // GroupNorm `n`'s gain folded into the stored tensor (a thread's items share the channel octet: 8 cached loads)
            float vals[8];
            unpack8(mv, vals);
            const f32x4 g0 = *(const f32x4*)(a.out_gain + cg), g1 = *(const f32x4*)(a.out_gain + cg + 4);
            vals[0] *= g0.x; vals[1] *= g0.y; vals[2] *= g0.z; vals[3] *= g0.w; vals[4] *= g1.x; vals[5] *= g1.y; vals[6] *= g1.z; vals[7] *= g1.w;
            mv = pack8(vals);
          }
          if (chs_out_p) {
            float q[8];
            unpack8(mv, q);
#pragma unroll
            for (int k = 0; k < 8; ++k) { c1[k] += q[k]; c2[k] = fmaf(q[k], q[k], c2[k]); }
          }
        }
        *(u32x4*)(a.y + off) = mv;
      }
    }
    // ---- seams: the tile's last row (-> the pooled row below) and last column (-> the pooled column to the right), one 16-byte piece per
    // thread each: seam_r [f][C/32][H/16][W][32], seam_c [f][C/32][W/16][H][32] (whole image rows / columns, so the corner pixel needs no case)
    {
      const int oct = tid & 15, q = tid >> 4;     // q = position along the seam
      const int cg = nt * 128 + oct * 8;
      if (cg < a.Cout) {
        if (ty0 + 16 < a.H) {
          const u32x4 v = *(const u32x4*)(smem + (15 * 16 + q) * PT_RS + oct * 16);
          *(u32x4*)(seam_r_p + ((size_t)((f * CB_out + (cg >> 5)) * tilesY + ty) * a.W + tx0 + q) * 32 + (cg & 31)) = v;
        }
        if (tx0 + 16 < a.W) {
          const u32x4 v = *(const u32x4*)(smem + (q * 16 + 15) * PT_RS + oct * 16);
          *(u32x4*)(seam_c_p + ((size_t)((f * CB_out + (cg >> 5)) * tilesX + tx) * a.H + ty0 + q) * 32 + (cg & 31)) = v;
        }
      }
    }
    s_sum2.x = p_sum; s_sum2.y = 0.f; s_sq2.x = p_sq; s_sq2.y = 0.f;
    if (chs_out_p) {
      // threads tid = octet + 16 q share an octet: lanes octet + 16 {0..3} of each wave (two exchanges), then the four waves through the
      // epilogue-table area of the LDS (dead since phase 1), then one fp64 atomic per (channel, moment): 256 per tile
      float* scr = (float*)(smem + KK_O);          // [4 waves][16 octets][16]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c1[k] += __shfl_xor(c1[k], 16, 64); c1[k] += __shfl_xor(c1[k], 32, 64);
        c2[k] += __shfl_xor(c2[k], 16, 64); c2[k] += __shfl_xor(c2[k], 32, 64);
      }
      if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { scr[(w * 16 + lane) * 16 + k] = c1[k]; scr[(w * 16 + lane) * 16 + 8 + k] = c2[k]; }
      }
      __syncthreads();
      {
        const int o = tid >> 4, k = tid & 15;
        const float t = (scr[(0 * 16 + o) * 16 + k] + scr[(1 * 16 + o) * 16 + k]) + (scr[(2 * 16 + o) * 16 + k] + scr[(3 * 16 + o) * 16 + k]);
        const int ch = nt * 128 + o * 8 + (k & 7);
        if (ch < a.Cout) atomicAdd(chs_out_p + ((size_t)f * a.Cout + ch) * 2 + (k >> 3), (double)t);
      }
    }
  } else {
  if (nvalid[1]) epilogue(std::integral_constant<int, 2>{});
  else if (nvalid[0]) epilogue(std::integral_constant<int, 1>{});
  }
  float s_sum = s_sum2.x + s_sum2.y, s_sq = s_sq2.x + s_sq2.y;
  if (GATE) {      // one fp64 atomic per tile: this tile's share of sum rstd0 dy xin
    float* red = (float*)(smem + KK_O + KK_BYTES);
    s_sum = wave_sum(s_sum + s_sq);
    if (lane == 0) red[w] = s_sum;
    __syncthreads();
    if (tid == 0) atomicAdd(a.gate_u + f, (double)((red[0] + red[1]) + (red[2] + red[3])));
  }
  if (!BWD && a.stats_out) {
    float* red = (float*)(smem + KK_O + KK_BYTES);
    s_sum = wave_sum(s_sum);
    s_sq = wave_sum(s_sq);
    if (lane == 0) { red[w] = s_sum; red[NWAVE + w] = s_sq; }
    __syncthreads();
    if (tid == 0) {
      float t1 = (red[0] + red[1]) + (red[2] + red[3]), t2 = (red[NWAVE] + red[NWAVE + 1]) + (red[NWAVE + 2] + red[NWAVE + 3]);
      if (NWAVE == 8) {
        t1 += (red[4] + red[5]) + (red[6] + red[7]);
        t2 += (red[12] + red[13]) + (red[14] + red[15]);
      }
      atomicAdd(a.stats_out + 2 * f, (double)t1);
      atomicAdd(a.stats_out + 2 * f + 1, (double)t2);
    }
  }
  if (TRACE && tid == 0) {
    long long* t = a.trace + (size_t)blockIdx.x * 12;
    t[0] = t_trace[0]; t[1] = t_trace[1]; t[2] = t_trace[2]; t[3] = wall_clock64(); t[4] = cu_key; t[5] = 0;
    for (int k = 0; k < 5; ++k) t[6 + k] = 0;   // (slots of the former per-subtile stamps: they perturbed the epilogue's waits)
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Small-batch (acting path, agent.py:190-206: B = 1 .. 8 frames) instantiation: the SAME convolution, layouts and arithmetic, tiled
// for LATENCY.  One frame of the 2x model gives the throughput kernel above 16 / 8 / 2 workgroups per layer, each walking the whole
// K = 1152 ... 2304 chain alone (26-28 us per launch on 2-32 of 256 CUs: profiles/r02_t1_step_kernel_stats.csv).  Here a workgroup
// takes 16 x 16 pixels x 32 output channels (4x the workgroups) with EIGHT waves -- wave w = pixel rows 2w, 2w+1 = one 2 x 16-pixel
// subtile, 18 MFMAs per wave and channel block: an eighth of the throughput kernel's serial chain, and two waves per SIMD to hide
// each other's LDS round trips.  Halo and the 18 KB weight slice of a channel block
// are double-buffered in LDS (one workgroup per CU is plenty), one barrier per channel block, plain epilogue.  Same K order
// (channel block, kernel row, kernel column, 16-channel half) as the throughput kernel.
#define S_W_BYTES (9 * 32 * 64)                       // 18432: 9 taps x 32 couts x 32 cin
#define S_NBUF 3                                      // stages in LDS: the loads of channel block cb + 2 are in flight while cb computes
#define S_KK_OFF (S_NBUF * (A_BYTES + S_W_BYTES))     // 133056
#define S_BYTES (S_KK_OFF + 9 * 32 * 4 + 64)

template <bool HAS_RES>
__global__ __launch_bounds__(512, 1) void vpt_conv3x3_small_kernel(VptConv3x3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[S_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int tilesX = a.W >> 4, tilesY = a.H >> 4;
  const int CB_out = a.Cout >> 5;
  int L = blockIdx.x;
  const int cbo = L % CB_out; L /= CB_out;
  const int tx = L % tilesX; L /= tilesX;
  const int ty = L % tilesY;
  const int f = L / tilesY;
  const int nt = cbo >> 2, qo = cbo & 3;
  const int tx0 = tx * 16, ty0 = ty * 16;
  const int NCB = a.Cin >> 5;
  const int HW = a.H * a.W;

  int a_loff[3];
  unsigned a_gbyte[3];
  unsigned a_inside = 0;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int q = tid + 512 * m;
    const int P = ((q >> 5) << 3) + (q & 7), part = (q >> 3) & 3;
    a_loff[m] = -1;
    a_gbyte[m] = 0u;
    if (P < 324) {
      const int hy = P / 18, hx = P - hy * 18;
      const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
      a_loff[m] = P * A_RS + part * 16;
      if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
        a_gbyte[m] = (unsigned)((y * a.W + x) * 32 + part * 8) * 2u;
        a_inside |= 1u << m;
      }
    }
  }
  const op16_t* xplane = a.x + (size_t)f * NCB * HW * 32;
  // weight slice of one channel block: 18 pieces of 1 KB (tap, 16-row half); wave w moves pieces w, w + 8, w + 16 (3 for waves 0 / 1,
  // 2 for the others).  A stage = those pieces by LDS-DMA + the halo's three 16-byte loads per lane into one of two register sets.
  const op16_t* wbase = a.wpk + (size_t)nt * NCB * 9 * 4096 + (size_t)(qo * 32) * 32 + (size_t)lane * 8;
  u32x4 areg[2][3];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  unsigned char* const wlds = smem + S_NBUF * A_BYTES;
#define S_ISSUE(cb_, SET_)                                                                                \
  do {                                                                                                    \
    unsigned char* wd_ = wlds + ((cb_) % S_NBUF) * S_W_BYTES;                                             \
    _Pragma("unroll") for (int p_ = w; p_ < 18; p_ += 8)                                                  \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + ((size_t)(cb_) * 9 + (p_ >> 1)) * 4096 + (p_ & 1) * 512), \
                                       (__attribute__((address_space(3))) void*)(wd_ + (p_ >> 1) * 2048 + (p_ & 1) * 1024), 16, 0, 0); \
    _Pragma("unroll") for (int m_ = 0; m_ < 3; ++m_)                                                      \
      areg[SET_][m_] = *(const u32x4*)((const char*)xplane + ((size_t)(cb_) * HW * 64 + a_gbyte[m_]));    \
  } while (0)
#define S_WRITE_A(cb_, SET_)                                                                              \
  _Pragma("unroll") for (int m_ = 0; m_ < 3; ++m_)                                                        \
    if (a_loff[m_] >= 0) *(u32x4*)(smem + ((cb_) % S_NBUF) * A_BYTES + a_loff[m_]) = ((a_inside >> m_) & 1u) ? areg[SET_][m_] : zero4
  // counted wait: everything up to and including stage `older` has landed, the stage issued after it (if any) stays in flight.
  // Loads retire in issue order; a stage is 6 vector-memory instructions for waves 0 / 1 and 5 for the others.
#define S_WAIT_OLDER(YOUNGER_ISSUED)                                                                      \
  do {                                                                                                    \
    if (!(YOUNGER_ISSUED)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               \
    else if (w < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                      \
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                                 \
  } while (0)

  S_ISSUE(0, 0);
  if (NCB > 1) S_ISSUE(1, 1);
  float mean, rstd;
  frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
  {
    float* kk = (float*)(smem + S_KK_OFF);
    for (int idx = tid; idx < 9 * 32; idx += 512) {
      const int o = (idx >> 5) * a.CoutPad + cbo * 32 + (idx & 31);
      kk[idx] = a.edge_sa[o] - rstd * mean * a.edge_sg[o];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the statistics / table loads above were issued after the stages: drain once)
  S_WRITE_A(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int bsw = (l31 >> 2) & 3;
  // iteration cb: issue stage cb + 2 (its halo into the register set stage cb used), compute stage cb, then move the halo of stage
  // cb + 1 -- issued one iteration EARLIER, so it has had two compute phases to land; the counted wait leaves stage cb + 2 in flight
  // -- from its registers into LDS for the next iteration.  One barrier per channel block.
  op16x8 sfb[3], sfa[3];
#define SB() __builtin_amdgcn_sched_barrier(0)
#define S_LOADG(g__)                                                                                      \
  do {                                                                                                    \
    const int gg_ = (g__);   /* a constant after unrolling: the register-set index gg_ % 3 must fold */ \
    const int tap_ = gg_ >> 1, ks_ = gg_ & 1, dy_ = tap_ / 3, dx_ = tap_ - 3 * dy_;                       \
    sfb[gg_ % 3] = *(const op16x8*)(bL + tap_ * 2048 + (((2 * ks_ + hi) ^ bsw) << 4));                    \
    sfa[gg_ % 3] = *(const op16x8*)(aL + (dy_ * 18 + dx_) * A_RS + ks_ * 32);                             \
  } while (0)
#define S_ITER(cb_, SET_)                                                                                 \
  do {                                                                                                    \
    const bool more_ = (cb_) + 2 < NCB;                                                                   \
    if (more_) S_ISSUE((cb_) + 2, SET_);                                                                  \
    const unsigned char* aL = smem + ((cb_) % S_NBUF) * A_BYTES + ((w * 2 + sub_row(l31)) * 18 + (l31 & 15)) * A_RS + hi * 16; \
    const unsigned char* bL = wlds + ((cb_) % S_NBUF) * S_W_BYTES + l31 * 64;                             \
    /* 18 groups (kernel row, kernel column, 16-channel half) of one weight fragment, one pixel fragment, one MFMA: the fragments \
       of group g + 2 are requested before the MFMA of group g issues (three register sets, order pinned; left alone the compiler \
       reads each group right before its use and waits for it). */ \
    S_LOADG(0); S_LOADG(1); SB();                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 18; ++g_) {                                                   \
      if (g_ + 2 < 18) { S_LOADG(g_ + 2); }                                                               \
      /* LDS reads return in order: group g_ has landed once at most the two younger groups' four reads are outstanding */ \
      if (g_ < 16) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                     \
      else if (g_ == 16) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                               \
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
      SB();                                                                                               \
      acc = VPT_MFMA_32X32X16(sfb[g_ % 3], sfa[g_ % 3], acc, 0, 0, 0);                                    \
      SB();                                                                                               \
    }                                                                                                     \
    if ((cb_) + 1 < NCB) {                                                                                \
      S_WAIT_OLDER(more_);                                                                                \
      S_WRITE_A((cb_) + 1, 1 - (SET_));                                                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
    }                                                                                                     \
    __builtin_amdgcn_s_barrier();                                                                         \
    asm volatile("" ::: "memory");                                                                        \
  } while (0)
  for (int cb = 0; cb < NCB; cb += 2) {
    S_ITER(cb, 0);
    if (cb + 1 < NCB) S_ITER(cb + 1, 1);
  }
#undef S_ITER
#undef S_LOADG
#undef SB
#undef S_ISSUE
#undef S_WRITE_A
#undef S_WAIT_OLDER

  // ---- epilogue: GroupNorm fold + ReLU (+ residual), 8-byte pieces (4 consecutive output channels of one pixel), statistics ----
  const float* kk = (const float*)(smem + S_KK_OFF);
  float s_sum = 0.f, s_sq = 0.f;
  const size_t obase = (size_t)(f * CB_out + cbo) * HW * 32;
  {
    const int y = ty0 + w * 2 + sub_row(l31);
    const int x = tx0 + (l31 & 15);
    const int ey = (y == 0) ? 0 : ((y == a.H - 1) ? 2 : 1);
    const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
    const float* ke = kk + (ey * 3 + ex) * 32 + 4 * hi;
    const size_t poff = obase + (size_t)(y * a.W + x) * 32 + 4 * hi;
    u32x2 r2[4];
    if (HAS_RES) {
#pragma unroll
      for (int g = 0; g < 4; ++g) r2[g] = *(const u32x2*)(a.res + poff + 8 * g);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 k4 = *(const f32x4*)(ke + 8 * g);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(rstd, acc[4 * g + j], k4[j]), 0.f);
      if (HAS_RES) {
        v[0] += op16_lo_to_f32(r2[g].x); v[1] += op16_hi_to_f32(r2[g].x);
        v[2] += op16_lo_to_f32(r2[g].y); v[3] += op16_hi_to_f32(r2[g].y);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { s_sum += v[j]; s_sq = fmaf(v[j], v[j], s_sq); }
      const u32x2 pk = {pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3])};
      *(u32x2*)(a.y + poff + 8 * g) = pk;
    }
  }
  if (a.stats_out) {
    float* red = (float*)(smem + S_KK_OFF + 9 * 32 * 4);
    s_sum = wave_sum(s_sum);
    s_sq = wave_sum(s_sq);
    if (lane == 0) { red[w] = s_sum; red[8 + w] = s_sq; }
    __syncthreads();
    if (tid == 0) {
      atomicAdd(a.stats_out + 2 * f, (double)(((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))));
      atomicAdd(a.stats_out + 2 * f + 1, (double)(((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]))));
    }
  }
}

static long long* g_conv_trace = nullptr;
extern "C" void vpt_conv3x3_set_trace(void* buf) { g_conv_trace = (long long*)buf; }  // profiling: [grid][12] int64, or null

extern "C" int vpt_conv3x3_launch(const VptConv3x3Args* a_in, hipStream_t stream) {
  int ablate = 0, extra_lds = 0;
#ifdef VPT_CONV_PROFILE
  static int ablate_env = -1, extra_lds_env = 0;
  if (ablate_env < 0) {
    const char* e = getenv("VPT_CONV_ABLATE");
    ablate_env = e ? atoi(e) : 0;
    const char* xl = getenv("VPT_CONV_EXTRA_LDS");  // dynamic LDS bytes (> 2 KB forces one workgroup per CU)
    extra_lds_env = xl ? atoi(xl) : 0;
    if (extra_lds_env > 0) {
      (void)hipFuncSetAttribute((const void*)vpt_conv3x3_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, extra_lds_env);
      (void)hipFuncSetAttribute((const void*)vpt_conv3x3_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, extra_lds_env);
    }
  }
  ablate = ablate_env; extra_lds = extra_lds_env;
#endif
  VptConv3x3Args a_copy = *a_in;
  a_copy.ablate = ablate;
  a_copy.trace = g_conv_trace;
  const VptConv3x3Args* a = &a_copy;
  if ((a->H & 15) || (a->W & 15) || (a->Cin & 31) || (a->Cout & 31) || a->frames <= 0) return -1;
  const long grid = (long)a->frames * (a->H >> 4) * (a->W >> 4) * a->NT;
  if (grid > 0x7fffffffL) return -2;
  const int mode = a->bwd ? (a->res ? 3 : (a->gate_stats ? 6 : 2)) : (a->pool ? (a->pool_mask ? 7 : 4) : (a->res ? (a->res_bias ? 5 : 1) : 0));
  if (a->pool_mask && !a->pool) return -1;
  if (a->gate_stats && (!a->bwd || a->res || !a->gate_u || a->trace)) return -1;   // the gated dgrad: no skip connection (a block's conv1 -> conv0)
  if ((a->res_bias != nullptr) != (a->res_scale != nullptr) || (a->res_bias && (a->bwd || !a->res))) return -1;
  if ((a->kk_frame != nullptr) != (a->rs_frame != nullptr) || (a->kk_frame && a->bwd)) return -1;
  if (a->bwd && (!a->xin || !a->coef)) return -1;   // dgrad always carries the GroupNorm-statistics terms (c0 + c1 * xin)
  if (a->pool && (a->bwd || a->res || (a->tiling != 1 && a->tiling != 3) || a->trace || !a->seam_r || !a->seam_c)) return -1;
  // the latency tiling (32 output channels per workgroup) is the CALLER's choice, never the grid size's: a frame's result must not
  // depend on how many frames share the launch (the two tilings sum a tile's statistics in different orders)
  if (!a->bwd && a->tiling != 1 && a->tiling != 2 && a->tiling != 3) return -1;
  if (!a->bwd && !a->trace && a->tiling == 2) {
    const long sgrid = (long)a->frames * (a->H >> 4) * (a->W >> 4) * (a->Cout >> 5);
    if (a->res) hipLaunchKernelGGL((vpt_conv3x3_small_kernel<true>), dim3((unsigned)sgrid), dim3(512), 0, stream, *a);
    else hipLaunchKernelGGL((vpt_conv3x3_small_kernel<false>), dim3((unsigned)sgrid), dim3(512), 0, stream, *a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
#define LAUNCH_(T_, M_) hipLaunchKernelGGL((vpt_conv3x3_kernel<T_, M_>), dim3((unsigned)grid), dim3(256), extra_lds, stream, *a)
  if (a->trace && mode > 3) return -1;
  if (mode == 6) { LAUNCH_(false, 6); return hipGetLastError() == hipSuccess ? 0 : -3; }
  if (mode == 7) { LAUNCH_(false, 7); return hipGetLastError() == hipSuccess ? 0 : -3; }
  // tiling 3 (forward) = the 32-row, eight-wave tiles wherever the image has whole 32-row bands.  Measured at parity with the 16-row tiles on
  // every layer shape (profiles/r04_experiments.md section 7: halving the weight DMA buys nothing on a power-limited chip), so the shipped
  // choice stays the 16-row kernel; the variant is kept selectable because it is bit-identical per pixel and covered by the same tests.
  if (!a->trace && !a->bwd && mode != 4 && (a->H & 31) == 0 && a->tiling == 3 && !extra_lds) {
    const long g32 = grid >> 1;
#define LAUNCH32_(M_) hipLaunchKernelGGL((vpt_conv3x3_kernel<false, M_, 32>), dim3((unsigned)g32), dim3(512), 0, stream, *a)
    if (mode == 0) LAUNCH32_(0); else if (mode == 1) LAUNCH32_(1); else LAUNCH32_(5);
#undef LAUNCH32_
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  if (a->trace) { if (mode == 0) LAUNCH_(true, 0); else if (mode == 1) LAUNCH_(true, 1); else if (mode == 2) LAUNCH_(true, 2); else LAUNCH_(true, 3); }
  else { if (mode == 0) LAUNCH_(false, 0); else if (mode == 1) LAUNCH_(false, 1); else if (mode == 2) LAUNCH_(false, 2); else if (mode == 3) LAUNCH_(false, 3); else if (mode == 4) LAUNCH_(false, 4); else LAUNCH_(false, 5); }
#undef LAUNCH_
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Seam pass of the pool-fused convolution (mode 4 above): the pooled pixels on a tile's first pooled row / column got the in-tile part of
// their 3 x 3 window only.  Pooled pixel (J, I), J % 8 == 0, J > 0: conv row 2J - 1 is row 15 of the tile row above = seam_r[J / 8 - 1],
// columns 2I - 1 .. 2I + 1 (whole image rows: the corner needs no case); I % 8 == 0, I > 0: conv column 2I - 1 = seam_c[I / 8 - 1], rows
// 2J - 1 .. 2J + 1.  A pixel on both kinds of seam is handled once, by its row item.  Adds these pixels' share of the frame statistics.
// Work per frame: ((H/16 - 1) * W/2 + (W/16 - 1) * (H/2 - (H/16 - 1))) pixels x C/8 sixteen-byte items -- 18 % of the pooled tensor at
// 64 x 64, instead of the whole pre-pool tensor written and read back.
template <bool MASK>      // MASK: the training forward's arg-max masks are finished too (the inference instantiation carries none of that code)
__global__ __launch_bounds__(256) void vpt_pool_seam_kernel(VptPoolSeamArgs a) {
  // one workgroup per (frame, 32-channel block): thread = (seam pixel slot tid >> 2, channel octet tid & 3), a slot walks the seam pixels in steps of 64
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  __shared__ float red[4][4][18];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int PH = a.H >> 1, PW = a.W >> 1, TY = a.H >> 4, TX = a.W >> 4;
  const int n_row = (TY - 1) * PW;                   // pixels on row seams
  const int col_len = PH - (TY - 1);                 // pixels of one column seam that are not on a row seam
  const int n_pix = n_row + (TX - 1) * col_len;
  const int cb = blockIdx.x % a.CB, f = blockIdx.x / a.CB;
  const int oct = tid & 3;
  const size_t plane = (size_t)(f * a.CB + cb);
  float s_sum = 0.f, s_sq = 0.f;                     // statistics of the UNscaled finished pixels (the frame statistics of P)
  float c1[8], c2[8];                                // per-channel sums of the STORED pixels
  float gv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { c1[k] = 0.f; c2[k] = 0.f; gv[k] = a.gain ? a.gain[cb * 32 + oct * 8 + k] : 1.f; }
  for (int pix = tid >> 2; pix < n_pix; pix += 64) {
    int J, I;
    if (pix < n_row) { J = (pix / PW + 1) * 8; I = pix % PW; }
    else {
      const int q = pix - n_row, t = q / col_len, k = q - t * col_len;   // k-th pooled row that is not a multiple of 8 (row 0 counts: it is not a seam)
      I = (t + 1) * 8;
      J = (k == 0) ? 0 : (k - 1) / 7 * 8 + (k - 1) % 7 + 1;
    }
    vpt_op16* yp = a.y + (plane * PH * PW + (size_t)(J * PW + I)) * 32 + oct * 8;
    i16x8 m = *(const i16x8*)yp;
    const i16x8 m_in = m;                      // the in-tile part of the maximum (what the convolution's epilogue stored)
    i16x8 sv[6];                               // the window positions outside the tile: row above (k = 0, 1, 2), column to the left (k = 0, 3, 6)
    bool have[6] = {false, false, false, false, false, false};
    if ((J & 7) == 0 && J > 0) {
      const vpt_op16* sr = a.seam_r + ((plane * TY + (J >> 3) - 1) * a.W) * 32 + oct * 8;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int x = 2 * I + dx;
        if (x >= 0) {
          const i16x8 v = *(const i16x8*)(sr + (size_t)x * 32);
          m = __builtin_elementwise_max(m, v);
          if (MASK) { sv[dx + 1] = v; have[dx + 1] = true; }
        }
      }
    }
    if ((I & 7) == 0 && I > 0) {
      const vpt_op16* sc = a.seam_c + ((plane * TX + (I >> 3) - 1) * a.H) * 32 + oct * 8;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int y = 2 * J + dy;
        if (y >= 0) {
          const i16x8 v = *(const i16x8*)(sc + (size_t)y * 32);
          m = __builtin_elementwise_max(m, v);
          if (MASK) { sv[3 + dy + 1] = v; have[3 + dy + 1] = true; }
        }
      }
    }
    if constexpr (MASK) {
      // arg-max masks of the training forward (vpt_conv3x3_kernel mode 7: bit 8 - k set = position k differs from the maximum): the in-tile bits
      // stay valid only where the in-tile maximum IS the window's maximum; the positions this kernel adds get their bits here
      typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
      unsigned short* mp = (unsigned short*)a.mask + (plane * PH * PW + (size_t)(J * PW + I)) * 32 + oct * 8;
      u16x8 mk = *(const u16x8*)mp;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        unsigned bits = (m_in[c] == m[c]) ? (unsigned)mk[c] : 0x1ffu;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int k = q < 3 ? q : 3 * (q - 3);            // scan position of seam value q
          if (have[q] && sv[q][c] == m[c]) bits &= ~(1u << (8 - k));
        }
        mk[c] = (unsigned short)bits;
      }
      *(u16x8*)mp = mk;
    }
    float vals[8];
    u32x4 mv = __builtin_bit_cast(u32x4, m);
    unpack8(mv, vals);
#pragma unroll
    for (int k = 0; k < 8; ++k) { s_sum += vals[k]; s_sq = fmaf(vals[k], vals[k], s_sq); }
    if (a.gain) {
#pragma unroll
      for (int k = 0; k < 8; ++k) vals[k] *= gv[k];
      mv = pack8(vals);
      unpack8(mv, vals);
    }
    *(u32x4*)yp = mv;
#pragma unroll
    for (int k = 0; k < 8; ++k) { c1[k] += vals[k]; c2[k] = fmaf(vals[k], vals[k], c2[k]); }
  }
  // reduce: lanes with the same octet (lane & 3), then the four waves
#pragma unroll
  for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { c1[k] += __shfl_xor(c1[k], off, 64); c2[k] += __shfl_xor(c2[k], off, 64); }
    s_sum += __shfl_xor(s_sum, off, 64); s_sq += __shfl_xor(s_sq, off, 64);
  }
  if (lane < 4) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[w][lane][k] = c1[k]; red[w][lane][8 + k] = c2[k]; }
    red[w][lane][16] = s_sum; red[w][lane][17] = s_sq;
  }
  __syncthreads();
  if (tid < 64 && a.chs_out) {
    const int o = tid >> 4, k = tid & 15;
    const float t = (red[0][o][k] + red[1][o][k]) + (red[2][o][k] + red[3][o][k]);
    atomicAdd(a.chs_out + ((size_t)f * a.CB * 32 + cb * 32 + o * 8 + (k & 7)) * 2 + (k >> 3), (double)t);
  }
  if (tid >= 64 && tid < 66 && a.stats_out) {        // the frame totals: four octets x four waves
    const int k = 16 + (tid - 64);
    float t = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) t += (red[0][o][k] + red[1][o][k]) + (red[2][o][k] + red[3][o][k]);
    atomicAdd(a.stats_out + 2 * f + (tid - 64), (double)t);
  }
}

extern "C" int vpt_pool_seam_launch(const VptPoolSeamArgs* a, hipStream_t stream) {
  if ((a->H & 15) || (a->W & 15) || a->frames <= 0 || a->CB <= 0) return -1;
  const int PH = a->H >> 1, PW = a->W >> 1, TY = a->H >> 4, TX = a->W >> 4;
  const int n_pix = (TY - 1) * PW + (TX - 1) * (PH - (TY - 1));
  if (n_pix == 0) return 0;                          // a single tile per frame: every window is inside it
  const long grid = (long)a->frames * a->CB;
  if (grid > 0x7fffffffL) return -2;
  if (a->mask) hipLaunchKernelGGL(vpt_pool_seam_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  else hipLaunchKernelGGL(vpt_pool_seam_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
