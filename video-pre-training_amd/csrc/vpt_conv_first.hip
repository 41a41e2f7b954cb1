// Stack-0 first convolution fused with the uint8 ingest, bias, ReLU and the 3x3/stride-2 max-pool (gfx950).
//
// Replaces: ImgPreprocessing.forward (x/255, lib/policy.py:39-45), the NHWC->NCHW permute of
// ImpalaCNN.forward (lib/impala_cnn.py:190), CnnDownStack.firstconv of stack 0 (Conv2d(3->C, 3x3, pad 1,
// bias) + ReLU, lib/impala_cnn.py:86-97 with lib/util.py:64-65) and F.max_pool2d(k3,s2,p1)
// (lib/impala_cnn.py:117).  The 128x128xC pre-pool activation (4 MB/frame at 2x) never reaches HBM.
//
// Formulation: K = 27 (+2 bias slots) padded to 32 -> two MFMA 32x32x16 k-steps with SWAPPED operands: the weights are the MFMA A
// operand (rows = output channels, resident in registers for the whole workgroup), the pixels are the B operand (0..255 are exact in
// bf16 and fp16; the 1/255 is folded into the packed weights; K slot order: vpt_conv_first_tile.h).  With the swap a lane ends up holding
// 16 output channels of ONE pixel, and the pool is lane-local (below).
//
// One workgroup = 8x8 pooled pixels (17x17 conv pixels, 19x19 input pixels) x 128 output channels, persistent over a tile range.
#include "vpt_common.h"
#include "vpt_kernels.h"
#include "vpt_conv_first_tile.h"

typedef int i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// ---- the pooled pixel's nine conv pixels are computed IN ITS LANE; the conv tile never exists (round 6) ------------------------------------------
// Until round 6 the kernel wrote the 17 x 17 x 128 conv tile to LDS (74 KB) and read it back nine times per pooled value (147 KB) -- the form
// vpt_conv_first_bwd_kernel still uses for its recompute (cf_conv_tile).  Here a wave owns 32 pooled pixels (4 rows x 8 columns; two such half-tiles
// one after the other) x 32 output channels and runs the conv of each of the window's nine positions with the pixels as MFMA columns: lane (hi, j)
// then holds the 16 channels {8 g + 4 hi + r} of position (dy, dx) of ITS pooled pixel and folds them into a running maximum that starts at 0 (= ReLU).
// 2 x the MFMAs (576 conv pixels computed for 289 distinct ones; the MFMA pipe was idle), no conv tile, no pooling pass, 29 KB of LDS instead of 80.
// What LDS still holds is the input: the 19 x 19 raw 8-byte records, converted ONCE per tile into 289 operand records of 64 bytes (the conv pixel's 32 K
// slots in fragment order: row 0 | row 1 | row 2 | ninth values, bias ones), so a fragment is two ds_read_b128.
// The 16-bit values are those of cf_conv_tile bit for bit (same MFMA, same operands, same k order; rounding is monotonic): the backward kernel's
// recompute finds its maxima.
// Measured (profiles/r06_experiments.md section 4, per 1024 frames, the inference call with gain + per-channel sums): LDS-tile kernel 0.577 ms ->
// 0.517.  Timing-only ablations (VPT_CF_ABLATE) put the rest at: output stores 0.07, the eight further positions 0.16 (the MFMA pipe 78 % busy while
// they run), conversion + first position 0.09, fetch / staging / barriers 0.15 -- the phases of a workgroup do not overlap, only workgroups do, which
// is why four 4-wave workgroups per CU beat two 8-wave ones (-6 %).
#ifndef VPT_CF_ABLATE
#define VPT_CF_ABLATE 0                   // timing builds only (tools/experiments/exp_r06_l.sh): 1 no output stores, 2 one window position instead of nine, 4 no statistics / gain / channel sums, 8 no conversion
#endif
#define OP_RS 80                          // record pitch: 64 bytes + 16 (consecutive records 20 banks apart: the conversion's ds_write_b128 are conflict-free)
#define OP_BYTES (289 * OP_RS)            // 23120
#define RAW_OFF OP_BYTES
#define SCR_OFF (RAW_OFF + 2896)          // 2888 bytes of raw records, padded to a multiple of 16
#define GAIN_OFF (SCR_OFF + 8 * 2 * 32 * 4)   // out_gain of the launch's NT x 128 channels (NT <= 2): read per tile with LDS latency instead of L2's
#define CF2_SMEM_BYTES (GAIN_OFF + 256 * 4)

#ifndef CF2_WAVES
#define CF2_WAVES 4                        // waves per workgroup: 4 = both half-tiles (4 pooled rows each) in every wave, 4 workgroups per CU; 8 = one half-tile per wave, 2 per CU (measured 6 % slower: fewer independent workgroups to overlap the phases)
#endif
#define CF2_THREADS (64 * CF2_WAVES)
#define CF2_NU (8 / CF2_WAVES)             // half-tiles per wave
template <bool CHS>
__global__ __launch_bounds__(CF2_THREADS, 4) void vpt_conv_first_kernel(VptConvFirstArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CF2_SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int cbl = CF2_WAVES == 8 ? wave >> 1 : wave;      // this wave's 32-channel block of the 128-channel tile
  const int unit0 = CF2_WAVES == 8 ? (wave & 1) : 0;      // its (first) half-tile
  const int pi = 4 * unit0 + (j >> 3), pj = j & 7;        // this lane's pooled pixel of the 8 x 8 tile (half-tile unit0)
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int tilesX = PW >> 3, tilesY = PH >> 3;
  const long T = (long)a.frames * tilesY * tilesX * a.NT;
  const int CB_out = a.Cout >> 5;

  u32x2 nxt[CF_FETCH(CF2_THREADS)];
  // (the thread index is made opaque per call in the three per-tile helpers: the record / pixel coordinates derived from it are a handful of
  // instructions to recompute and a dozen registers to keep across the tile loop -- the ones that spilled)
  auto fetch = [&](int f, int ty, int tx) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    cf_fetch_input<CF2_THREADS>(a.img + (size_t)f * a.H * a.W * 3, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2, t_, nxt);
  };
  auto stage = [&](int ty, int tx) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    cf_stage_input<CF2_THREADS>(smem + RAW_OFF, nxt, t_, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2);
  };
  // raw records -> operand records (thread t < 289 owns conv pixel t).  The 16-byte chunks of the records of every second PAIR of conv rows are stored
  // swapped two by two (^ 16): lanes of neighbouring pooled rows then read 4 banks apart instead of from the same ones
  auto convert = [&](bool top, bool left) {       // top / left: the tile touches the image's first row / column -- its conv row / column 0 is the pool's padding
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int m = 0; m < (289 + CF2_THREADS - 1) / CF2_THREADS; ++m) {
      const int t = t_ + CF2_THREADS * m;
      if (t < 289) {
        const int cr = t / 17, cc = t - cr * 17;
        const unsigned char* ib = smem + RAW_OFF + (cr * 19 + cc) * 8;
        const u32x2 r0 = *(const u32x2*)ib, r1 = *(const u32x2*)(ib + 19 * 8), r2 = *(const u32x2*)(ib + 2 * 19 * 8);
        const float n0 = (float)ib[2 * 8 + 2], n1 = (float)ib[(19 + 2) * 8 + 2], n2 = (float)ib[(2 * 19 + 2) * 8 + 2];
        u32x4 ex;
        ex.x = pack_op16x2_exact(n0, n1);
        ex.y = pack_op16x2_exact(n2, 1.0f);
        ex.z = CF_ONE_BITS;
        ex.w = 0u;
        unsigned char* dst = smem + t * OP_RS;
        const int swz = ((cr >> 1) & 1) << 4;
        // a conv pixel outside the image gets the all-zero record (bias slots too): its conv result is exactly 0 and never beats the running maximum
        const uint32_t keep = ((top && cr == 0) || (left && cc == 0)) ? 0u : 0xffffffffu;
        const u32x4 k4 = {keep, keep, keep, keep};
        *(u32x4*)(dst + (0 ^ swz)) = cf_bytes8(r0) & k4;
        *(u32x4*)(dst + (16 ^ swz)) = cf_bytes8(r1) & k4;
        *(u32x4*)(dst + (32 ^ swz)) = cf_bytes8(r2) & k4;
        *(u32x4*)(dst + (48 ^ swz)) = ex & k4;
      }
    }
  };
  // Contiguous tile range per workgroup: consecutive tiles belong to the same frame, so the frame statistics are summed in registers and flushed with
  // one atomic pair per (wave, frame).  CHS: the running per-channel sums are fp32 over a GROUP of tiles (a quarter of a 128 x 128 frame's 64) and fp64
  // across groups; the workgroup ranges are cut at group boundaries only, so which tiles share an fp32 sum never depends on the number of frames in
  // the launch -- a frame's statistics, hence its result, must not depend on how the batch is chunked.  Tile coordinates are
  // decoded once and then counted up (the 64-bit divisions of a per-tile decode were ~1000 scalar instructions per tile).
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

  if (a.out_gain && tid < 256) ((float*)(smem + GAIN_OFF))[tid] = tid < a.Cout ? a.out_gain[tid] : 0.f;
  if (t_begin < t_end) {
    fetch(f, ty, tx);
    stage(ty, tx);
  }
  __syncthreads();
  if (t_begin < t_end) convert(ty == 0, tx == 0);
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
#pragma unroll
    for (int u = 0; u < CF2_NU; ++u) {
    if (valid) {
      const int rec0u = rec00 + u * (8 * 17 * OP_RS);      // the wave's second half-tile: pooled rows + 4 = conv rows + 8 (same row-pair parity)
      // Running maximum over the nine positions on the fp32 BIT PATTERNS as signed integers, starting from 0: a positive float's pattern orders like
      // an integer, every negative one (and -0) is a negative integer and loses against the 0 -- max-pool and ReLU in one, two positions per
      // v_max3_i32.  Rounding to 16 bits is monotonic, so rounding the maximum once gives the maximum of the rounded values bit for bit.
      int mi[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) mi[r] = 0;
      auto conv_at = [&](int pos) -> f32x16 {
        const int dy = pos / 3, dx = pos - 3 * dy;
        const unsigned char* fp = smem + rec0u + (dy * 17 + dx) * OP_RS + (dy == 2 ? ch2 : ch01);
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
      const int pi_ = 4 * (unit0 + u) + ((lane_ & 31) >> 3), pj_ = lane_ & 7;
      off = ((size_t)(f * CB_out + (cg >> 5)) * PH * PW + (size_t)((py0 + pi_) * PW + px0 + pj_)) * 32 + hi_ * 16;
    }
    if (u + 1 < CF2_NU && valid && (!(VPT_CF_ABLATE & 1) || va.x == 0x12345678u)) {      // (all but the wave's last half-tile: stored at once)
      *(u32x4*)(a.y + off) = va;
      *(u32x4*)(a.y + off + 8) = vb;
    }
    }
    // staging FIRST, stores after it: gfx950 counts loads and stores in one counter, and the staging waits for the fetch with vmcnt(0) -- behind the
    // stores that would be this tile's write latency in front of the barrier, in every tile
    if (tile + 1 < t_end) stage(nty, ntx);   // the raw area's last readers (this tile's conversion) finished before the barrier that ended the previous iteration
    if (valid && (!(VPT_CF_ABLATE & 1) || va.x == 0x12345678u)) {
      *(u32x4*)(a.y + off) = va;
      *(u32x4*)(a.y + off + 8) = vb;
    }
    __syncthreads();                    // every fragment read of this tile done; raw records of the next one visible
    if (tile + 1 < t_end && !(VPT_CF_ABLATE & 8)) convert(nty == 0, ntx == 0);
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
      const float t = CF2_WAVES == 8 ? scr[((2 * cb) * 2 + h) * 32 + mo * 16 + k] + scr[((2 * cb + 1) * 2 + h) * 32 + mo * 16 + k]
                                     : scr[(cb * 2 + h) * 32 + mo * 16 + k];
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
  if (a->out_gain && a->Cout > 256) return -1;                      // the kernel keeps the launch's gains in 1 KB of LDS (the model's widest stack 0 is 192 channels)
  if (a->chs_out && a->NT != 1) return -1;                          // running per-channel sums: one channel tile (Cout <= 128); else vpt_channel_stats
  const long per_cu = 16 / CF2_WAVES;
  if (grid > per_cu * num_cu) grid = per_cu * num_cu;
  if (a->chs_out) hipLaunchKernelGGL(vpt_conv_first_kernel<true>, dim3((unsigned)grid), dim3(CF2_THREADS), 0, stream, *a);
  else hipLaunchKernelGGL(vpt_conv_first_kernel<false>, dim3((unsigned)grid), dim3(CF2_THREADS), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
