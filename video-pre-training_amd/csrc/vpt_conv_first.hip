// Stack-0 first convolution fused with the uint8 ingest, bias, ReLU and the 3x3/stride-2 max-pool (gfx950).
//
// Replaces: ImgPreprocessing.forward (x/255, lib/policy.py:39-45), the NHWC->NCHW permute of
// ImpalaCNN.forward (lib/impala_cnn.py:190), CnnDownStack.firstconv of stack 0 (Conv2d(3->C, 3x3, pad 1,
// bias) + ReLU, lib/impala_cnn.py:86-97 with lib/util.py:64-65) and F.max_pool2d(k3,s2,p1)
// (lib/impala_cnn.py:117).  The 128x128xC pre-pool activation (4 MB/frame at 2x) never reaches HBM.
//
// Formulation: K = 27 (+2 bias slots) padded to 32 -> two MFMA 32x32x16 k-steps with SWAPPED operands:
// the weights are the MFMA A operand (rows = output channels, resident in registers for the whole
// workgroup), the pixels are the B operand: the input tile is converted ONCE to 16-bit operands in LDS (0..255 are
// exact in bf16 and fp16; the 1/255 is folded into the packed weights) and the K slots are ordered so that a lane's
// fragment is one 16-byte LDS read (vpt_conv_first_tile.h).  With the swap each lane ends up holding 4 consecutive
// output channels of one pixel, so the conv tile is written to LDS with packed 8-byte stores, and the
// pool is a packed signed-16-bit max over the raw bf16 bit patterns starting from 0, which is max-pool and
// ReLU in one (positive bf16 patterns order like integers; negative ones are negative integers).
//
// One workgroup = 8x8 pooled pixels (17x17 conv pixels, 19x19 input pixels) x 128 output channels.
#include "vpt_common.h"
#include "vpt_kernels.h"
#include "vpt_conv_first_tile.h"
#include <stdlib.h>

typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Eight waves per workgroup, two workgroups per CU (the 79 KB conv tile decides that): four waves per SIMD.  With four waves per
// workgroup (round 2) a tile took ~11 k cycles against ~5 k of issued work -- LDS round trips, the slice counter and three
// barriers per tile with nothing else to run.
#define CF_THREADS 512

// CHS: also accumulate the per-channel sums of the stored tensor (a.chs_out, NT = 1).  The pooling items are then mapped so that BOTH items
// of a thread have the same channel octet (tid & 15): 16 running sums per thread instead of 32 -- the kernel must stay within 128 registers
// (four waves per SIMD) -- and the 16 lanes of a ds_read_b128 group read the 256 contiguous bytes of one conv pixel.
template <bool CHS>
__global__ __launch_bounds__(CF_THREADS, 4) void vpt_conv_first_lds_kernel(VptConvFirstArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CF_SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int tilesX = PW >> 3, tilesY = PH >> 3;
  const long T = (long)a.frames * tilesY * tilesX * a.NT;
  const int CB_out = a.Cout >> 5;

  // persistent workgroups (2 per CU): the next tile's 19 x 19 x 3 input bytes are fetched into registers while the
  // current tile computes, so the global-load latency is off the per-tile critical path
  u32x2 nxt[CF_FETCH(CF_THREADS)];
  auto fetch = [&](int f, int ty, int tx) {
    cf_fetch_input<CF_THREADS>(a.img + (size_t)f * a.H * a.W * 3, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2, tid, nxt);
  };
  // contiguous tile range per workgroup: consecutive tiles belong to the same frame, so the frame statistics are
  // summed in registers and flushed with one atomic pair per (workgroup, frame)
  // CHS: the running per-channel sums are fp32 over a GROUP of tiles (a quarter of a 128 x 128 frame's 64) and fp64 across groups; the
  // workgroup ranges are cut at group boundaries only, so which tiles share an fp32 sum never depends on the number of frames in the
  // launch -- a frame's statistics, hence its result, must not depend on how the batch is chunked
  const int tiles_per_frame = tilesY * tilesX * a.NT;
  const int group = !CHS ? 1 : ((tiles_per_frame & 15) == 0 ? 16 : tiles_per_frame);
  const long n_groups = (T + group - 1) / group;
  const long per = ((n_groups + gridDim.x - 1) / gridDim.x) * group;
  const long t_begin = blockIdx.x * per, t_end = min(t_begin + per, T);
  // tile coordinates (frame, tile row, tile column, channel tile) are decoded once and then counted up: the 64-bit
  // divisions of a per-tile decode were ~1000 scalar instructions per tile
  int nt, tx, ty, f;
  {
    long L = t_begin;
    nt = (int)(L % a.NT); L /= a.NT;
    tx = (int)(L % tilesX); L /= tilesX;
    ty = (int)(L % tilesY);
    f = (int)(L / tilesY);
  }
  int nnt = nt, ntx = tx, nty = ty, nf = f;   // the tile after the current one
  int in_group = 0;                           // tiles of the current group done (t_begin is a multiple of `group`: no 64-bit modulo per tile)
  auto advance = [&]() {
    if (++nnt == a.NT) { nnt = 0; if (++ntx == tilesX) { ntx = 0; if (++nty == tilesY) { nty = 0; ++nf; } } }
  };
  // (per tile the sums are fp32 in a fixed order; ACROSS tiles they are added in fp64, so a frame's statistics do not
  // depend on how the tile list happens to be cut into workgroup ranges, i.e. on the batch size)
  int nt_loaded = -1, stat_f = -1;
  double d_sum = 0.0, d_sq = 0.0;
  // Per-channel sums of the STORED tensor (a.chs_out: the GroupNorm-`n` fold needs sum_p Q and sum_p Q^2 per channel and frame).  A thread's
  // pooling items keep their channel octet from tile to tile, so the sums run in registers over a group of 16 tiles and are combined
  // across threads (4 lanes x 8 waves per octet) once per group, through the then idle conv-tile area of the LDS, ending in one fp64
  // atomic per (channel, moment).
  float c1[8], c2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { c1[k] = 0.f; c2[k] = 0.f; }
  auto flush_channel_sums = [&](int fr) {      // uniform; the caller guarantees the conv tile is idle and follows up with a barrier
    float* scr = (float*)smem;                 // [8 waves][16 octets][16]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      c1[k] += __shfl_xor(c1[k], 16, 64); c1[k] += __shfl_xor(c1[k], 32, 64);
      c2[k] += __shfl_xor(c2[k], 16, 64); c2[k] += __shfl_xor(c2[k], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { scr[((tid >> 6) * 16 + lane) * 16 + k] = c1[k]; scr[((tid >> 6) * 16 + lane) * 16 + 8 + k] = c2[k]; }
    }
    __syncthreads();
    if (tid < 256) {
      const int o = tid >> 4, k = tid & 15;            // channel octet, value (0..7 sums, 8..15 sums of squares)
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += scr[(w * 16 + o) * 16 + k];
      const int ch = o * 8 + (k & 7);
      if (ch < a.Cout) atomicAdd(a.chs_out + ((size_t)fr * a.Cout + ch) * 2 + (k >> 3), (double)t);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { c1[k] = 0.f; c2[k] = 0.f; }
  };
  op16x8 wfr[4][2];
  // Two barriers per tile: [records(t) staged, counter 0] -> fetch(t + 1) into registers, conv slices -> barrier -> stage
  // records(t + 1) (the slices were their last readers), reset the counter, pool the conv tile -> barrier.
  if (t_begin < t_end) {
    fetch(f, ty, tx);
    cf_stage_input<CF_THREADS>(smem + IN_OFF, nxt, tid, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2);
    if (tid == 0) *(int*)(smem + CTR_OFF) = 0;
  }
  __syncthreads();
  for (long tile = t_begin; tile < t_end; ++tile, nt = nnt, tx = ntx, ty = nty, f = nf) {
    const int py0 = ty * 8, px0 = tx * 8;
    advance();
    if (nt != nt_loaded) {   // weight fragments [nt][cs][ks][lane][8] stay in registers across tiles
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wfr[cs][ks] = *((const op16x8*)a.wfrag + ((nt * 4 + cs) * 2 + ks) * 64 + lane);
      nt_loaded = nt;
    }
    if (tile + 1 < t_end) fetch(nf, nty, ntx);
    cf_conv_tile(smem, wfr, lane, py0, px0, ty == 0 || tx == 0);
    __syncthreads();
    if (tile + 1 < t_end) cf_stage_input<CF_THREADS>(smem + IN_OFF, nxt, tid, a.H, a.W, 2 * (nty * 8) - 2, 2 * (ntx * 8) - 2);
    if (tid == 0) *(int*)(smem + CTR_OFF) = 0;

  // ---- 3x3 / stride 2 max-pool over the conv tile, store + statistics ----
  if (a.stats_out && f != stat_f) {
    if (stat_f >= 0) {
      const double t1 = wave_sum_f64(d_sum), t2 = wave_sum_f64(d_sq);
      if (lane == 0) {
        atomicAdd(a.stats_out + 2 * stat_f, t1);
        atomicAdd(a.stats_out + 2 * stat_f + 1, t2);
      }
    }
    stat_f = f; d_sum = 0.0; d_sq = 0.0;
  }
  float s_sum = 0.f, s_sq = 0.f;
#pragma unroll
  for (int it = 0; it < 1024 / CF_THREADS; ++it) {
    const int item = tid + CF_THREADS * it;
    // A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (+32): the four lane quads of a
    // group take pooled columns 0, 2, 4, 6 or 1, 3, 5, 7 -- neighbouring pooled pixels are 2 * CT_RS = 8 banks apart and a quad
    // covers 16, so consecutive columns collide two by two (576 conflict cycles per tile, profiles/r03_experiments.md section 8)
    int oct4, pxl, pyl, cbl;
    if (CHS) { oct4 = item & 3; cbl = (item >> 2) & 3; pxl = (item >> 4) & 7; pyl = item >> 7; }     // octet (cbl, oct4) = tid & 15 for both items
    else { oct4 = item & 3; pxl = (0x76452310u >> (4 * ((item >> 2) & 7))) & 7; pyl = (item >> 5) & 7; cbl = item >> 8; }
    const int cg = nt * 128 + cbl * 32 + oct4 * 8;
    const unsigned char* src = smem + ((2 * pyl) * 17 + 2 * pxl) * CT_RS + (cbl * 32 + oct4 * 8) * 2;
    i16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};    // = ReLU: positive bf16 patterns order like signed 16-bit integers, negative ones stay below 0
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const i16x8 v = *(const i16x8*)(src + (dy * 17 + dx) * CT_RS);
        m = __builtin_elementwise_max(m, v);
      }
    if (cg < a.Cout) {
      u32x4 mv = __builtin_bit_cast(u32x4, m);
      const uint32_t ones = CF_ONE_BITS | (CF_ONE_BITS << 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {     // packed pairs: one dot2 per two values for the sum, one for the sum of squares (products of 16-bit operands are exact in fp32)
        s_sum = dot2_op16(mv[k], ones, s_sum);
        s_sq = dot2_op16(mv[k], mv[k], s_sq);
      }
      if (a.out_gain) {   // GroupNorm `n`'s gain folded into the stored tensor (the statistics above are those of the unscaled values)
        const f32x4 g0 = *(const f32x4*)(a.out_gain + cg), g1 = *(const f32x4*)(a.out_gain + cg + 4);
        float vals[8];
        unpack8(mv, vals);
        vals[0] *= g0.x; vals[1] *= g0.y; vals[2] *= g0.z; vals[3] *= g0.w; vals[4] *= g1.x; vals[5] *= g1.y; vals[6] *= g1.z; vals[7] *= g1.w;
        mv = pack8(vals);
      }
      if (CHS) {          // per-channel sums of what is STORED (rounded to 16 bits, scaled)
        float q[8];
        unpack8(mv, q);
#pragma unroll
        for (int k = 0; k < 8; ++k) { c1[k] += q[k]; c2[k] = fmaf(q[k], q[k], c2[k]); }
      }
      const size_t off = ((size_t)(f * CB_out + (cg >> 5)) * PH * PW + (size_t)((py0 + pyl) * PW + px0 + pxl)) * 32 + (cg & 31);
      *(u32x4*)(a.y + off) = mv;
    }
  }
    if (a.stats_out) {   // per LANE in fp64 across tiles; the lanes are combined when the frame changes (a wave reduction per tile cost 0.09 of this kernel's 0.57 ms per 1024 frames)
      d_sum += (double)s_sum;
      d_sq += (double)s_sq;
    }
    __syncthreads();   // all pooling reads of the conv tile done before the next tile overwrites it
    if (CHS && (++in_group == group || tile + 1 >= t_end)) {         // last tile of a group (groups never straddle frames): hand the per-channel sums over
      in_group = 0;
      flush_channel_sums(f);
      __syncthreads();
    }
  }
  if (a.stats_out && stat_f >= 0) {
    const double t1 = wave_sum_f64(d_sum), t2 = wave_sum_f64(d_sq);
    if (lane == 0) {
      atomicAdd(a.stats_out + 2 * stat_f, t1);
      atomicAdd(a.stats_out + 2 * stat_f + 1, t2);
    }
  }
}


// ---- round 6: the pooled pixel's nine conv pixels are computed IN ITS LANE; the conv tile never exists ------------------------------------------
// The kernel above writes the 17 x 17 x 128 conv tile to LDS (74 KB) and reads it back nine times per pooled value (147 KB): the LDS pipe bounds it
// (61 % busy, a third of that in conflict stalls; 5.4 k cycles per tile and CU against 0.64 k of MFMA).  Here a wave owns 32 pooled pixels (4 rows x 8
// columns) x 32 output channels and runs the conv of each of the window's nine positions with the pixels as MFMA columns: lane (hi, j) then holds the
// 16 channels {8 g + 4 hi + r} of position (dy, dx) of ITS pooled pixel, rounds them to 16 bits and folds them into a running packed signed-16-bit
// maximum that starts at 0 (= ReLU, as above).  2 x the MFMAs (576 conv pixels computed for 289 distinct ones -- the MFMA pipe was idle), no conv tile, no
// pooling pass.  What LDS still holds is the input: the 19 x 19 raw 8-byte records as before, converted ONCE per tile into 289 operand records of 64
// bytes (the conv pixel's 32 K slots in fragment order: row 0 | row 1 | row 2 | ninth values, bias ones), so a fragment is two ds_read_b128.
// The 16-bit values are those of the kernel above bit for bit (same MFMA, same operands, same k order): vpt_conv_first_bwd_kernel's recompute, which still
// goes through cf_conv_tile, finds its maxima.
#ifndef VPT_CF_ABLATE
#define VPT_CF_ABLATE 0                   // timing builds only (tools/experiments/exp_r06_l.sh): 1 no output stores, 2 one window position instead of nine, 4 no statistics / gain / channel sums, 8 no conversion
#endif
#define OP_RS 80                          // record pitch: 64 bytes + 16 (consecutive records 20 banks apart: the conversion's ds_write_b128 are conflict-free)
#define OP_BYTES (289 * OP_RS)            // 23120
#define ZERO_OFF OP_BYTES                 // one all-zero record: the fragment of a conv pixel outside the image (bias slot 0 too: the conv result is exactly 0)
#define RAW_OFF (ZERO_OFF + 64)
#define SCR_OFF (RAW_OFF + 2896)          // 2888 bytes of raw records, padded to a multiple of 16
#define GAIN_OFF (SCR_OFF + 8 * 2 * 32 * 4)   // out_gain of the launch's NT x 128 channels (NT <= 2): read per tile with LDS latency instead of L2's
#define CF2_SMEM_BYTES (GAIN_OFF + 256 * 4)

template <bool CHS>
__global__ __launch_bounds__(CF_THREADS, 4) void vpt_conv_first_kernel(VptConvFirstArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CF2_SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int cbl = wave >> 1;                              // this wave's 32-channel block of the 128-channel tile
  const int pi = 4 * (wave & 1) + (j >> 3), pj = j & 7;   // this lane's pooled pixel of the 8 x 8 tile
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int tilesX = PW >> 3, tilesY = PH >> 3;
  const long T = (long)a.frames * tilesY * tilesX * a.NT;
  const int CB_out = a.Cout >> 5;

  u32x2 nxt[1];
  // (the thread index is made opaque per call in the three per-tile helpers: the record / pixel coordinates derived from it are a handful of
  // instructions to recompute and a dozen registers to keep across the tile loop -- the ones that spilled)
  auto fetch = [&](int f, int ty, int tx) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    cf_fetch_input<CF_THREADS>(a.img + (size_t)f * a.H * a.W * 3, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2, t_, nxt);
  };
  auto stage = [&](int ty, int tx) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    cf_stage_input<CF_THREADS>(smem + RAW_OFF, nxt, t_, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2);
  };
  // raw records -> operand records (thread t < 289 owns conv pixel t).  The 16-byte chunks of the records of every second PAIR of conv rows are stored
  // swapped two by two (^ 16): lanes of neighbouring pooled rows then read 4 banks apart instead of from the same ones
  auto convert = [&]() {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (t_ < 289) {
      const int cr = t_ / 17, cc = t_ - cr * 17;
      const unsigned char* ib = smem + RAW_OFF + (cr * 19 + cc) * 8;
      const u32x2 r0 = *(const u32x2*)ib, r1 = *(const u32x2*)(ib + 19 * 8), r2 = *(const u32x2*)(ib + 2 * 19 * 8);
      const float n0 = (float)ib[2 * 8 + 2], n1 = (float)ib[(19 + 2) * 8 + 2], n2 = (float)ib[(2 * 19 + 2) * 8 + 2];
      u32x4 ex;
      ex.x = pack_op16x2_exact(n0, n1);
      ex.y = pack_op16x2_exact(n2, 1.0f);
      ex.z = CF_ONE_BITS;
      ex.w = 0u;
      unsigned char* dst = smem + t_ * OP_RS;
      const int swz = ((cr >> 1) & 1) << 4;
      *(u32x4*)(dst + (0 ^ swz)) = cf_bytes8(r0);
      *(u32x4*)(dst + (16 ^ swz)) = cf_bytes8(r1);
      *(u32x4*)(dst + (32 ^ swz)) = cf_bytes8(r2);
      *(u32x4*)(dst + (48 ^ swz)) = ex;
    }
  };
  // tile enumeration, statistics grouping: exactly as in the kernel above (a frame's statistics must not depend on how the batch is chunked)
  const int tiles_per_frame = tilesY * tilesX * a.NT;
  const int group = !CHS ? 1 : ((tiles_per_frame & 15) == 0 ? 16 : tiles_per_frame);
  const long n_groups = (T + group - 1) / group;
  const long per = ((n_groups + gridDim.x - 1) / gridDim.x) * group;
  const long t_begin = blockIdx.x * per, t_end = min(t_begin + per, T);
  int nt, tx, ty, f;
  {
    long L = t_begin;
    nt = (int)(L % a.NT); L /= a.NT;
    tx = (int)(L % tilesX); L /= tilesX;
    ty = (int)(L % tilesY);
    f = (int)(L / tilesY);
  }
  int nnt = nt, ntx = tx, nty = ty, nf = f;
  int in_group = 0;
  auto advance = [&]() {
    if (++nnt == a.NT) { nnt = 0; if (++ntx == tilesX) { ntx = 0; if (++nty == tilesY) { nty = 0; ++nf; } } }
  };
  int nt_loaded = -1, stat_f = -1;
  double d_sum = 0.0, d_sq = 0.0;
  // per-channel sums of the STORED tensor: a lane's 16 channels are the same in every tile, so the sums run in registers over a group of tiles; they
  // are combined across the 32 pixels of a half-wave by shuffles and across the two waves of a channel block through 2 KB of LDS, once per group
  f32x2 c1[CHS ? 8 : 1], c2[CHS ? 8 : 1];          // [dword k of the lane's 16 channels]: channels 2 k, 2 k + 1
#pragma unroll
  for (int k = 0; k < (CHS ? 8 : 1); ++k) { c1[k] = (f32x2){0.f, 0.f}; c2[k] = (f32x2){0.f, 0.f}; }
  float* scr = (float*)(smem + SCR_OFF);       // [8 waves][2 halves][16 sums | 16 sums of squares]
  op16x8 wfr[2];
  // lane constants of the fragment reads: record of the window's position (0, 0), chunk offsets (the ^ 16 swizzle follows the conv row PAIR: rows 2 pi
  // and 2 pi + 1 are pair pi, row 2 pi + 2 is pair pi + 1)
  const int rec00 = ((2 * pi) * 17 + 2 * pj) * OP_RS;
  const int ch01 = (hi << 4) ^ ((pi & 1) << 4), ch2 = (hi << 4) ^ (((pi + 1) & 1) << 4);

  if (tid < 16) *(uint32_t*)(smem + ZERO_OFF + tid * 4) = 0u;
  if (a.out_gain && tid < 256) ((float*)(smem + GAIN_OFF))[tid] = tid < a.Cout ? a.out_gain[tid] : 0.f;
  if (t_begin < t_end) {
    fetch(f, ty, tx);
    stage(ty, tx);
  }
  __syncthreads();
  if (t_begin < t_end) convert();
  __syncthreads();
  for (long tile = t_begin; tile < t_end; ++tile, nt = nnt, tx = ntx, ty = nty, f = nf) {
    const int py0 = ty * 8, px0 = tx * 8;
    advance();
    const int cg = nt * 128 + cbl * 32;
    const bool valid = cg < a.Cout;              // wave-uniform
    int lane_ = lane;                            // opaque per tile: the per-lane POINTERS built from it (weights, gains, output) are two registers
    asm volatile("" : "+v"(lane_));              // each to keep across the loop and one or two instructions to rebuild
    const int hi_ = lane_ >> 5;
    if (nt != nt_loaded) {
      u32x4 wq[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wq[ks] = *((const u32x4*)a.wfrag + ((nt * 4 + cbl) * 2 + ks) * 64 + lane_);
      // (opaque use: the s_waitcnt for these loads belongs HERE -- left to the first MFMA it is a vmcnt(0) behind the next tile's input fetch in
      // every tile, i.e. a global round trip in front of the compute)
      asm volatile("" : "+v"(wq[0]), "+v"(wq[1]));
      wfr[0] = __builtin_bit_cast(op16x8, wq[0]);
      wfr[1] = __builtin_bit_cast(op16x8, wq[1]);
      nt_loaded = nt;
    }
    if (tile + 1 < t_end) fetch(nf, nty, ntx);
    if (a.stats_out && f != stat_f) {
      if (stat_f >= 0) {
        const double t1 = wave_sum_f64(d_sum), t2 = wave_sum_f64(d_sq);
        if (lane == 0) {
          atomicAdd(a.stats_out + 2 * stat_f, t1);
          atomicAdd(a.stats_out + 2 * stat_f + 1, t2);
        }
      }
      stat_f = f; d_sum = 0.0; d_sq = 0.0;
    }
    u32x4 va, vb;
    size_t off;
    if (valid) {
      // conv pixels outside the image (the pool's padding row / column: tiles on the top / left border) read the zero record
      const bool out_t = ty == 0 && pi == 0, out_l = tx == 0 && pj == 0;
      // Running maximum over the nine positions on the fp32 BIT PATTERNS as signed integers, starting from 0: a positive float's pattern orders like
      // an integer, every negative one (and -0) is a negative integer and loses against the 0 -- max-pool and ReLU in one, two positions per
      // v_max3_i32.  Rounding to 16 bits is monotonic, so rounding the maximum once gives the maximum of the rounded values bit for bit.
      int mi[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) mi[r] = 0;
      auto conv_at = [&](int pos) -> f32x16 {
        const int dy = pos / 3, dx = pos - 3 * dy;
        int rec = rec00 + (dy * 17 + dx) * OP_RS;
        if (dy == 0) rec = out_t ? ZERO_OFF : rec;
        if (dx == 0) rec = out_l ? ZERO_OFF : rec;
        const unsigned char* fp = smem + rec + (dy == 2 ? ch2 : ch01);
        const op16x8 p0 = *(const op16x8*)fp, p1 = *(const op16x8*)(fp + 32);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = VPT_MFMA_32X32X16(wfr[0], p0, acc, 0, 0, 0);
        return VPT_MFMA_32X32X16(wfr[1], p1, acc, 0, 0, 0);
      };
#pragma unroll
      for (int pp = 0; pp < ((VPT_CF_ABLATE & 2) ? 0 : 4); ++pp) {
        // (whole-vector casts: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0 whatever the index -- ROCm 7.2 clang)
        const i32x16 ca = __builtin_bit_cast(i32x16, conv_at(2 * pp)), cb = __builtin_bit_cast(i32x16, conv_at(2 * pp + 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) mi[r] = max(max(mi[r], ca[r]), cb[r]);
      }
      {
        const i32x16 ca = __builtin_bit_cast(i32x16, conv_at(8));
#pragma unroll
        for (int r = 0; r < 16; ++r) mi[r] = max(mi[r], ca[r]);
      }
      uint32_t m[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = pack_op16x2(__builtin_bit_cast(float, mi[2 * k]), __builtin_bit_cast(float, mi[2 * k + 1]));
      // m[2 g + d] = channels cg + 8 g + 4 hi + 2 d, + 1 of pooled pixel (py0 + pi, px0 + pj)
      float s_sum = 0.f, s_sq = 0.f;
      const uint32_t ones = CF_ONE_BITS | (CF_ONE_BITS << 16);
#pragma unroll
      for (int k = 0; k < 8; ++k) {     // statistics of the unscaled values (products of 16-bit operands are exact in fp32)
        s_sum = dot2_op16(m[k], ones, s_sum);
        s_sq = dot2_op16(m[k], m[k], s_sq);
      }
      if (a.stats_out && !(VPT_CF_ABLATE & 4)) {                // per LANE in fp64 across tiles; the lanes are combined when the frame changes
        d_sum += (double)s_sum;
        d_sq += (double)s_sq;
      }
      if (a.out_gain && !(VPT_CF_ABLATE & 4)) {                 // GroupNorm `n`'s gain folded into the stored tensor
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 gn = *(const f32x4*)((const float*)(smem + GAIN_OFF) + cg + 8 * g + 4 * hi_);
          const f32x2 q0 = (f32x2){op16_lo_to_f32(m[2 * g]), op16_hi_to_f32(m[2 * g])} * (f32x2){gn.x, gn.y};
          const f32x2 q1 = (f32x2){op16_lo_to_f32(m[2 * g + 1]), op16_hi_to_f32(m[2 * g + 1])} * (f32x2){gn.z, gn.w};
          m[2 * g] = pack_op16x2(q0.x, q0.y);
          m[2 * g + 1] = pack_op16x2(q1.x, q1.y);
        }
      }
      if (CHS && !(VPT_CF_ABLATE & 4)) {                        // per-channel sums of what is STORED (rounded to 16 bits, scaled): packed fp32 pairs
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const f32x2 q = {op16_lo_to_f32(m[k]), op16_hi_to_f32(m[k])};
          c1[k] += q;
          c2[k] = q * q + c2[k];
        }
      }
      // v_permlane32_swap: the lower half-wave ends up with channels 0..15 of its pixel's block, the upper one with 16..31 -- 32 contiguous bytes per lane
      const auto s0 = __builtin_amdgcn_permlane32_swap(m[0], m[4], false, false), s1 = __builtin_amdgcn_permlane32_swap(m[1], m[5], false, false);
      const auto s2 = __builtin_amdgcn_permlane32_swap(m[2], m[6], false, false), s3 = __builtin_amdgcn_permlane32_swap(m[3], m[7], false, false);
      va = (u32x4){s0[0], s1[0], s0[1], s1[1]};
      vb = (u32x4){s2[0], s3[0], s2[1], s3[1]};
      const int pi_ = 4 * (wave & 1) + ((lane_ & 31) >> 3), pj_ = lane_ & 7;
      off = ((size_t)(f * CB_out + (cg >> 5)) * PH * PW + (size_t)((py0 + pi_) * PW + px0 + pj_)) * 32 + hi_ * 16;
    }
    // staging FIRST, stores after it: gfx950 counts loads and stores in one counter, and the staging waits for the fetch with vmcnt(0) -- behind the
    // stores that would be this tile's write latency in front of the barrier, in every tile
    if (tile + 1 < t_end) stage(nty, ntx);   // the raw area's last readers (this tile's conversion) finished before the barrier that ended the previous iteration
    if (valid && (!(VPT_CF_ABLATE & 1) || va.x == 0x12345678u)) {
      *(u32x4*)(a.y + off) = va;
      *(u32x4*)(a.y + off + 8) = vb;
    }
    __syncthreads();                    // every fragment read of this tile done; raw records of the next one visible
    if (tile + 1 < t_end && !(VPT_CF_ABLATE & 8)) convert();
    const bool flush = CHS && (++in_group == group || tile + 1 >= t_end);      // last tile of a group (groups never straddle frames)
    if (flush) {
#pragma unroll
      for (int k = 0; k < (CHS ? 8 : 1); ++k) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          c1[k].x += __shfl_xor(c1[k].x, o, 64); c1[k].y += __shfl_xor(c1[k].y, o, 64);
          c2[k].x += __shfl_xor(c2[k].x, o, 64); c2[k].y += __shfl_xor(c2[k].y, o, 64);
        }
      }
      if (j == 0) {
#pragma unroll
        for (int k = 0; k < (CHS ? 8 : 1); ++k) {
          *(f32x2*)(scr + (wave * 2 + hi) * 32 + 2 * k) = c1[k];
          *(f32x2*)(scr + (wave * 2 + hi) * 32 + 16 + 2 * k) = c2[k];
        }
      }
#pragma unroll
      for (int k = 0; k < (CHS ? 8 : 1); ++k) { c1[k] = (f32x2){0.f, 0.f}; c2[k] = (f32x2){0.f, 0.f}; }
    }
    __syncthreads();                    // operand records of the next tile (and a flush's partial sums) visible
    if (flush && tid < 256) {           // (the scratch is next written a tile later at the earliest, behind two barriers)
      int t_ = tid;
      asm volatile("" : "+v"(t_));      // (opaque: no per-thread atomic address kept across the loop)
      const int mo = t_ & 1, k = (t_ >> 1) & 15, h = (t_ >> 5) & 1, cb = t_ >> 6;
      const float t = scr[((2 * cb) * 2 + h) * 32 + mo * 16 + k] + scr[((2 * cb + 1) * 2 + h) * 32 + mo * 16 + k];
      const int ch = cb * 32 + 8 * (k >> 2) + 4 * h + (k & 3);
      if (ch < a.Cout) atomicAdd(a.chs_out + ((size_t)f * a.Cout + ch) * 2 + mo, (double)t);
    }
    if (flush) in_group = 0;
  }
  if (a.stats_out && stat_f >= 0) {
    const double t1 = wave_sum_f64(d_sum), t2 = wave_sum_f64(d_sq);
    if (lane == 0) {
      atomicAdd(a.stats_out + 2 * stat_f, t1);
      atomicAdd(a.stats_out + 2 * stat_f + 1, t2);
    }
  }
}

extern "C" int vpt_conv_first_launch(const VptConvFirstArgs* a, hipStream_t stream) {
  if ((a->H & 15) || (a->W & 15) || (a->Cout & 31) || a->frames <= 0) return -1;
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                 ? prop.multiProcessorCount : 256;
  }
  long grid = (long)a->frames * (a->H >> 4) * (a->W >> 4) * a->NT;
  if ((long)a->frames * a->H * a->W * 3 > 0x7fffffffL) return -2;   // 32-bit pixel offsets inside a launch
  if (a->chs_out && a->NT != 1) return -1;                          // running per-channel sums: one channel tile (Cout <= 128); else vpt_channel_stats
  if (grid > 2L * num_cu) grid = 2L * num_cu;
  static const bool lds_tile = getenv("VPT_CONV_FIRST_LDS_TILE") != nullptr;      // A/B only (tools/experiments/exp_r06_k.sh): the conv-tile-in-LDS kernel
  if (lds_tile) {
    if (a->chs_out) hipLaunchKernelGGL(vpt_conv_first_lds_kernel<true>, dim3((unsigned)grid), dim3(CF_THREADS), 0, stream, *a);
    else hipLaunchKernelGGL(vpt_conv_first_lds_kernel<false>, dim3((unsigned)grid), dim3(CF_THREADS), 0, stream, *a);
  } else if (a->chs_out) hipLaunchKernelGGL(vpt_conv_first_kernel<true>, dim3((unsigned)grid), dim3(CF_THREADS), 0, stream, *a);
  else hipLaunchKernelGGL(vpt_conv_first_kernel<false>, dim3((unsigned)grid), dim3(CF_THREADS), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
