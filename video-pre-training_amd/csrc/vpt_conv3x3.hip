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

#define A_RS 80
#define A_BYTES (324 * A_RS)            // 25920
#define B_BYTES (3 * 128 * 64)          // 24576 per buffer (unpadded, swizzled)
#define KK_OFF (A_BYTES + 2 * B_BYTES)  // 75072
#define KK_BYTES (9 * 128 * 4)          // 4608
#define SMEM_BYTES (KK_OFF + KK_BYTES + 32)  // 79712 (+32: block stats reduction)
// epilogue staging (overlays the halo / weight buffers after the last step): per wave [res|out tile][xin tile], each
// 2 channel blocks x 32 pixels x 64 B at an 80-byte pixel pitch (16-byte aligned rows, <= 2-way conflicts on the 8-byte side)
#define ST_RS 80
#define ST_N2 (32 * ST_RS)              // 2560
#define ST_X (2 * ST_N2)                // 5120
#define ST_WAVE (2 * ST_X)              // 10240 per wave, 40960 per workgroup

// MFMA M-subtile row i (0..31) -> pixel (row 0/1, col 0..15) of a 2x16 patch.  Rows are swapped for columns
// 4..11 so that each 16-lane ds_read_b128 group {0-3,12-15,20-27} / {4-11,16-19,28-31} stays in one image row.
__device__ __forceinline__ int sub_row(int i) { return ((i >> 4) ^ (i >> 2) ^ (i >> 3)) & 1; }


// COUNTED: the wait in front of each step's barrier is a counted s_waitcnt that retires the weight DMA only and leaves the
// younger halo / residual prefetch loads in flight across the barrier (a plain __syncthreads() drains vmcnt to 0 because an
// LDS-DMA is pending).
template <bool COUNTED>
__global__ __launch_bounds__(256, 2) void vpt_conv3x3_kernel(VptConv3x3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  // profiling (vpt_conv3x3_set_trace): CU id + 100 MHz timestamps of the tile's phases
  int cu_key = -1;
  long long t_trace[3];
  if (a.trace && tid == 0) {
    t_trace[0] = wall_clock64();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID[3:0]
    cu_key = (int)(((xcc & 15u) << 8) | ((hw >> 8) & 0xffu));
  }
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int tilesX = a.W >> 4, tilesY = a.H >> 4;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = L % a.NT; L /= a.NT;
  const int tx = L % tilesX; L /= tilesX;
  const int ty = L % tilesY;
  const int f = L / tilesY;
  const int tx0 = tx * 16, ty0 = ty * 16;
  const int NCB = a.Cin >> 5;
  const int HW = a.H * a.W;

  // ---- halo staging map: chunk q -> (pixel P = 8*(q>>5) + (q&7), part = (q>>3)&3) ----
  int a_loff[6];            // LDS byte offset, -1: no chunk (beyond the 324 halo pixels)
  unsigned a_gbyte[6];      // byte offset inside one channel-block plane (clamped to 0 outside the image)
  unsigned a_inside = 0;    // bit m: chunk m lies inside the image (else literal zeros: the conv pads the NORMALISED tensor)
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int q = tid + 256 * m;
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
  // weight DMA: wave w moves pieces (4*m + w), m = 0..5, of the 24 KB step tile; lane = 16-byte chunk
  const op16_t* wbase = a.wpk + (size_t)nt * NCB * 9 * 4096 + (size_t)(w * 64 + lane) * 8;
  unsigned char* bdst = smem + A_BYTES + w * 1024;

#define ISSUE_B(step_, buf_)                                                                              \
  do {                                                                                                    \
    const op16_t* wp_ = wbase + (size_t)(step_) * 12288;                                                  \
    _Pragma("unroll") for (int m_ = 0; m_ < 6; ++m_)                                                      \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp_ + m_ * 2048),  \
                                       (__attribute__((address_space(3))) void*)(bdst + (buf_) * B_BYTES + m_ * 4096), \
                                       16, 0, 0);                                                         \
  } while (0)

  u32x4 areg[6];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- prologue: weights of step 0 (DMA), halo of channel block 0, epilogue constant table ----
  ISSUE_B(0, 0);
#pragma unroll
  for (int m = 0; m < 6; ++m) areg[m] = *(const u32x4*)((const char*)xplane + a_gbyte[m]);
  float mean = 0.f, rstd = 1.f, c0f = 0.f, c1f = 0.f;
  if (!a.bwd) {
    frame_mean_rstd(a.stats_in, f, a.inv_count_in, mean, rstd);
  } else if (a.coef) {
    c0f = a.coef[2 * f];
    c1f = a.coef[2 * f + 1];
  }
  {
    float* kk = (float*)(smem + KK_OFF);
    float ksa[5], ksg[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {  // 9*128 = 4.5 * 256 entries: all ten loads in flight together
      const int idx = tid + 256 * k;
      const int o = (idx >> 7) * a.CoutPad + nt * 128 + (idx & 127);
      ksa[k] = (idx < 9 * 128 && !a.bwd) ? a.edge_sa[o] : 0.f;
      ksg[k] = (idx < 9 * 128 && !a.bwd) ? a.edge_sg[o] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int idx = tid + 256 * k;
      if (idx < 9 * 128) kk[idx] = ksa[k] - rstd * mean * ksg[k];
    }
  }
#define WRITE_HALO()                                                                                      \
  _Pragma("unroll") for (int m_ = 0; m_ < 6; ++m_)                                                        \
    if (a_loff[m_] >= 0) *(u32x4*)(smem + a_loff[m_]) = ((a_inside >> m_) & 1u) ? areg[m_] : zero4
  WRITE_HALO();
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // fragment base addresses
  const unsigned char* aL = smem + ((wm * 8 + sub_row(l31)) * 18 + (l31 & 15)) * A_RS + hi * 16;
  const int bsw = (l31 >> 2) & 3;
  const unsigned char* bL0 = smem + A_BYTES + (wn * 64 + l31) * 64 + (((0 + hi) ^ bsw) << 4);  // ks = 0
  const unsigned char* bL1 = smem + A_BYTES + (wn * 64 + l31) * 64 + (((2 + hi) ^ bsw) << 4);  // ks = 1

  // epilogue addressing (needed early: the residual is requested during the last channel block)
  const int CB_out = a.Cout >> 5;
  const int cb0 = nt * 4 + wn * 2;                 // 32-channel block of n2 = 0
  const bool nvalid[2] = {(cb0 + 0) < CB_out, (cb0 + 1) < CB_out};
  const size_t nstep = (size_t)HW * 32;            // next 32-channel block
  // Coalesced side of the epilogue: global accesses are 16 bytes per lane over 16 consecutive pixels x 64 B (one full
  // 1 KB run per instruction) and pass through a wave-private LDS staging tile; the lane-local side (one pixel, four
  // consecutive channels per access) only touches LDS.  Going to memory straight from the accumulator layout made
  // every instruction visit 32 cache lines for 16 useful bytes each -- the L1 line rate, not HBM, was what the
  // residual / output traffic of the K = 1152 layers was waiting for.
  const int cq = lane >> 2, cchunk = lane & 3;         // staging role: pixel column of the 2x16 patch, 16-byte chunk
  int st_lds[2];                                      // LDS offset of (patch row j2, column cq) in l31 order
  size_t goff0[2];                                    // global element offset of (m = 0, j2) for channel block cb0
  const size_t gm = (size_t)(2 * a.W) * 32;           // + m * gm: two image rows further down
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2) {
    const int l31q = cq + 16 * ((j2 ^ (cq >> 2) ^ (cq >> 3)) & 1);   // inverse of sub_row(): lane that owns this pixel
    st_lds[j2] = l31q * ST_RS + cchunk * 16;
    const int y = ty0 + wm * 8 + j2, x = tx0 + cq;
    goff0[j2] = ((size_t)(f * CB_out + (nvalid[0] ? cb0 : 0)) * HW + (size_t)(y * a.W + x)) * 32 + cchunk * 8;
  }
  const int ll_lds = l31 * ST_RS + 8 * hi;             // lane-local: + n2 * ST_N2 + 16 * g
  unsigned char* stg = smem + w * ST_WAVE;             // [residual / output tile][xin tile], reused for every m
  u32x4 rq[4][2][2];
#define LOAD_RES(m_)                                                                                      \
  _Pragma("unroll") for (int n2_ = 0; n2_ < 2; ++n2_)                                                     \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                      \
      rq[m_][n2_][j_] = *(const u32x4*)(a.res + goff0[j_] + (m_) * gm + (nvalid[n2_] ? n2_ : 0) * nstep)

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
  const op16_t* resp = a.res ? a.res : a.y;   // no residual: the prefetch still runs (exact counted waits), result unused
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MM(set_, m_, n_) acc[m_][n_] = VPT_MFMA_32X32X16(fb[set_][n_], fa[set_][m_], acc[m_][n_], 0, 0, 0)
#define FA_LD(set_, dy_, g_, m_) \
  fa[set_][m_] = *(const op16x8*)(aL + ((dy_) * 18 + ((g_) >> 1)) * A_RS + (m_) * (2 * 18 * A_RS) + ((g_) & 1) * 32)
#define FB_LD(set_, g_, n_, boff_) \
  fb[set_][n_] = *(const op16x8*)((((g_) & 1) ? bL1 : bL0) + (boff_) + ((g_) >> 1) * (128 * 64) + (n_) * (32 * 64))
  // MFMAs of register set `set_`; fragments of group (ndy_, ng_) go to the other set; X0 / X1: the two free slots
#define GROUP(set_, ndy_, ng_, nboff_, X0, X1)                                                            \
  do {                                                                                                    \
    MM(set_, 0, 0); FB_LD(1 - (set_), ng_, 0, nboff_); FA_LD(1 - (set_), ndy_, ng_, 0); SB();             \
    MM(set_, 0, 1); FB_LD(1 - (set_), ng_, 1, nboff_); FA_LD(1 - (set_), ndy_, ng_, 1); SB();             \
    MM(set_, 1, 0); FA_LD(1 - (set_), ndy_, ng_, 2); FA_LD(1 - (set_), ndy_, ng_, 3); SB();               \
    MM(set_, 1, 1); SB();                                                                                 \
    MM(set_, 2, 0); X0; SB();                                                                             \
    MM(set_, 2, 1); SB();                                                                                 \
    MM(set_, 3, 0); X1; SB();                                                                             \
    MM(set_, 3, 1); SB();                                                                                 \
  } while (0)
#define GROUP_TAIL(set_)                                                                                  \
  do {                                                                                                    \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) { MM(set_, m_, 0); MM(set_, m_, 1); }                \
  } while (0)
#define GLDS(m_)                                                                                          \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp_ + (m_) * 2048),    \
                                   (__attribute__((address_space(3))) void*)(bd_ + (m_) * 4096), 16, 0, 0)
#define XA(m_) areg[m_] = *(const u32x4*)((const char*)xplane + (cbo_ + a_gbyte[m_]))
#define XR(m_, n2_, j_) rq[m_][n2_][j_] = *(const u32x4*)(resp + goff0[j_] + (m_) * gm + (nvalid[n2_] ? n2_ : 0) * nstep)
#define NOP_() ((void)0)
#define WAIT_BARRIER(n_late_)                                                                             \
  do {                                                                                                    \
    if constexpr (COUNTED) {                                                                              \
      asm volatile("s_waitcnt vmcnt(" #n_late_ ") lgkmcnt(0)" ::: "memory");                              \
      __builtin_amdgcn_s_barrier();                                                                       \
      asm volatile("" ::: "memory");                                                                      \
    } else {                                                                                              \
      __syncthreads();                                                                                    \
    }                                                                                                     \
    SB();                                                                                                 \
  } while (0)
  // Memory ops are issued in a FIXED order and count per wave -- weight DMA (6), then either the halo of the next block
  // (6, PRE_A) or the residual of subtiles 0 / 1 (8, PRE_R) -- so the counted wait in front of the barrier is exact: it
  // retires the DMA and leaves the younger loads in flight across the barrier.
#define CONV_STEP(cb_, dy_, NEXT_DY, PRE_A, WR_A, PRE_R, LAST)                                            \
  do {                                                                                                    \
    const int s_ = (cb_) * 3 + (dy_);                                                                     \
    const int boff_ = (s_ & 1) * B_BYTES, noff_ = B_BYTES - boff_;                                        \
    const op16_t* wp_ = wbase + (size_t)(s_ + 1) * 12288;                                                 \
    unsigned char* bd_ = bdst + noff_;                                                                    \
    const unsigned cbo_ = (unsigned)((cb_) + 1) * (unsigned)HW * 64u; /* bytes; uniform */                \
    if (LAST) {                                                                                           \
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
      WAIT_BARRIER(6);                                                                                    \
    } else if (PRE_R) {                                                                                   \
      GROUP(1, dy_, 4, boff_, do { XR(0, 0, 0); XR(0, 0, 1); } while (0), do { XR(0, 1, 0); XR(0, 1, 1); } while (0)); \
      GROUP(0, dy_, 5, boff_, do { XR(1, 0, 0); XR(1, 0, 1); } while (0), do { XR(1, 1, 0); XR(1, 1, 1); } while (0)); \
      WAIT_BARRIER(8);                                                                                    \
    } else {                                                                                              \
      GROUP(1, dy_, 4, boff_, NOP_(), NOP_());                                                            \
      GROUP(0, dy_, 5, boff_, NOP_(), NOP_());                                                            \
      WAIT_BARRIER(0);                                                                                    \
    }                                                                                                     \
    if (WR_A) {   /* the halo of the next channel block replaces the current one: second barrier before its first read */ \
      _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                                  \
        if (a_loff[m_] >= 0) *(u32x4*)(smem + a_loff[m_]) = ((a_inside >> m_) & 1u) ? areg[m_] : zero4;   \
        if (m_ + 4 < 6 && a_loff[m_ + 4] >= 0) *(u32x4*)(smem + a_loff[m_ + 4]) = ((a_inside >> (m_ + 4)) & 1u) ? areg[m_ + 4] : zero4; \
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

  if (a.trace && tid == 0) t_trace[1] = wall_clock64();
  if (a.ablate != 2) {
    FB_LD(0, 0, 0, 0); FA_LD(0, 0, 0, 0); FB_LD(0, 0, 1, 0); FA_LD(0, 0, 0, 1); FA_LD(0, 0, 0, 2); FA_LD(0, 0, 0, 3);
    SB();
    for (int cb = 0; cb + 1 < NCB; ++cb) {
      CONV_STEP(cb, 0, 1, true, false, false, false);
      CONV_STEP(cb, 1, 2, false, false, false, false);
      CONV_STEP(cb, 2, 0, false, true, false, false);
    }
    // last channel block: no further halo -> request the residual of the first two subtiles instead
    CONV_STEP(NCB - 1, 0, 1, false, false, true, false);
    CONV_STEP(NCB - 1, 1, 2, false, false, false, false);
    CONV_STEP(NCB - 1, 2, 0, false, false, false, true);
  } else {
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
#undef FA_LD
#undef FB_LD

  // ---------------- epilogue ----------------
  if (a.trace && tid == 0) t_trace[2] = wall_clock64();
  if (a.ablate == 1) {  // profiling: keep the accumulators live, skip the epilogue
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
  // the fragment registers are dead: the residual of subtiles 2 / 3 travels while subtiles 0 / 1 are processed
  if (a.res) {
    if (a.ablate == 2) { LOAD_RES(0); LOAD_RES(1); }
    LOAD_RES(2); LOAD_RES(3);
  }
  SB();

  // Operands are SWAPPED in the MFMA (weights = A rows, pixels = B columns), so a lane holds ONE pixel
  // (column l31 of the subtile) and, per accumulator, four groups of 4 consecutive output channels
  // (rows (r&3) + 8*(r>>2) + 4*hi): the arithmetic is lane-local -- 16-byte reads of the constant table, 8-byte
  // reads of the staged residual / xin, 8-byte writes of the bf16 result back into the staging tile -- and the
  // staging tile moves to / from memory in full 1 KB runs (see above).  The last step's barrier has retired every
  // read of the weight / halo buffers, so the staging tiles may overlay them.
  int eoff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int y = ty0 + wm * 8 + 2 * m + sub_row(l31);
    const int x = tx0 + (l31 & 15);
    const int ey = (y == 0) ? 0 : ((y == a.H - 1) ? 2 : 1);
    const int ex = (x == 0) ? 0 : ((x == a.W - 1) ? 2 : 1);
    eoff[m] = (ey * 3 + ex) * 128 + wn * 64 + 4 * hi;
  }
  const float* kk = (const float*)(smem + KK_OFF);
  float s_sum = 0.f, s_sq = 0.f;
  const bool use_x = a.bwd && a.xin;
  u32x4 xq[2][2];
  if (use_x) {
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
      for (int j = 0; j < 2; ++j) xq[n2][j] = *(const u32x4*)(a.xin + goff0[j] + (nvalid[n2] ? n2 : 0) * nstep);
  }

#pragma unroll
  for (int m = 0; m < 4; ++m) {
    // stage the coalesced inputs of this 2x16-pixel subtile
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (a.res) *(u32x4*)(stg + n2 * ST_N2 + st_lds[j]) = rq[m][n2][j];
        if (use_x) *(u32x4*)(stg + ST_X + n2 * ST_N2 + st_lds[j]) = xq[n2][j];
      }
    if (use_x && m < 3) {   // next subtile's xin: in flight while this one is processed
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int j = 0; j < 2; ++j) xq[n2][j] = *(const u32x4*)(a.xin + goff0[j] + (m + 1) * gm + (nvalid[n2] ? n2 : 0) * nstep);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: in-order LDS queue, no barrier needed
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      if (!nvalid[n2]) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 k4 = *(const f32x4*)(kk + eoff[m] + n2 * 32 + 8 * g);
        float v0 = fmaf(rstd, acc[m][n2][4 * g + 0], k4.x), v1 = fmaf(rstd, acc[m][n2][4 * g + 1], k4.y);
        float v2 = fmaf(rstd, acc[m][n2][4 * g + 2], k4.z), v3 = fmaf(rstd, acc[m][n2][4 * g + 3], k4.w);
        unsigned char* cell = stg + n2 * ST_N2 + ll_lds + 16 * g;
        if (!a.bwd) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        } else if (use_x) {  // dgrad: + d(mu, rstd)/dx terms of the GroupNorm statistics
          const u32x2 xi = *(const u32x2*)(cell + ST_X);
          v0 += fmaf(c1f, op16_lo_to_f32(xi.x), c0f); v1 += fmaf(c1f, op16_hi_to_f32(xi.x), c0f);
          v2 += fmaf(c1f, op16_lo_to_f32(xi.y), c0f); v3 += fmaf(c1f, op16_hi_to_f32(xi.y), c0f);
        }
        if (a.res) {
          const u32x2 r2 = *(const u32x2*)cell;
          v0 += op16_lo_to_f32(r2.x); v1 += op16_hi_to_f32(r2.x);
          v2 += op16_lo_to_f32(r2.y); v3 += op16_hi_to_f32(r2.y);
        }
        s_sum += (v0 + v1) + (v2 + v3);
        s_sq = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, s_sq))));
        const u32x2 pk = {pack_op16x2(v0, v1), pack_op16x2(v2, v3)};
        *(u32x2*)cell = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      if (!nvalid[n2]) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *(u32x4*)(a.y + goff0[j] + m * gm + n2 * nstep) = *(const u32x4*)(stg + n2 * ST_N2 + st_lds[j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is rewritten by the next subtile
  }
  if (a.stats_out) {
    float* red = (float*)(smem + KK_OFF + KK_BYTES);
    s_sum = wave_sum(s_sum);
    s_sq = wave_sum(s_sq);
    if (lane == 0) { red[w] = s_sum; red[4 + w] = s_sq; }
    __syncthreads();
    if (tid == 0) {
      atomicAdd(a.stats_out + 2 * f, (double)((red[0] + red[1]) + (red[2] + red[3])));
      atomicAdd(a.stats_out + 2 * f + 1, (double)((red[4] + red[5]) + (red[6] + red[7])));
    }
  }
  if (a.trace && tid == 0) {
    long long* t = a.trace + (size_t)blockIdx.x * 6;
    t[0] = t_trace[0]; t[1] = t_trace[1]; t[2] = t_trace[2]; t[3] = wall_clock64(); t[4] = cu_key; t[5] = 0;
  }
}

static long long* g_conv_trace = nullptr;
extern "C" void vpt_conv3x3_set_trace(void* buf) { g_conv_trace = (long long*)buf; }  // profiling: [grid][6] int64, or null

extern "C" int vpt_conv3x3_launch(const VptConv3x3Args* a_in, hipStream_t stream) {
  static int ablate = -1, extra_lds = 0;
  if (ablate < 0) {
    const char* e = getenv("VPT_CONV_ABLATE");
    ablate = e ? atoi(e) : 0;
    const char* xl = getenv("VPT_CONV_EXTRA_LDS");  // profiling: dynamic LDS bytes (> 2 KB forces one workgroup per CU)
    extra_lds = xl ? atoi(xl) : 0;
    if (extra_lds > 0) {
      hipFuncSetAttribute((const void*)vpt_conv3x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, extra_lds);
      hipFuncSetAttribute((const void*)vpt_conv3x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, extra_lds);
    }
  }
  VptConv3x3Args a_copy = *a_in;
  a_copy.ablate = ablate;
  a_copy.trace = g_conv_trace;
  const VptConv3x3Args* a = &a_copy;
  if ((a->H & 15) || (a->W & 15) || (a->Cin & 31) || (a->Cout & 31) || a->frames <= 0) return -1;
  const long grid = (long)a->frames * (a->H >> 4) * (a->W >> 4) * a->NT;
  if (grid > 0x7fffffffL) return -2;
  static int counted = -1;
  if (counted < 0) { const char* e = getenv("VPT_CONV_COUNTED"); counted = e ? atoi(e) : 1; }
  if (counted) hipLaunchKernelGGL(vpt_conv3x3_kernel<true>, dim3((unsigned)grid), dim3(256), extra_lds, stream, *a);
  else hipLaunchKernelGGL(vpt_conv3x3_kernel<false>, dim3((unsigned)grid), dim3(256), extra_lds, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
