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

typedef short i16x8 __attribute__((ext_vector_type(8)));
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
__global__ __launch_bounds__(CF_THREADS, 4) void vpt_conv_first_kernel(VptConvFirstArgs a) {
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
    cf_stage_input<CF_THREADS>(smem, nxt, tid);
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
    if (tile + 1 < t_end) cf_stage_input<CF_THREADS>(smem, nxt, tid);
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
  if (a->chs_out) hipLaunchKernelGGL(vpt_conv_first_kernel<true>, dim3((unsigned)grid), dim3(CF_THREADS), 0, stream, *a);
  else hipLaunchKernelGGL(vpt_conv_first_kernel<false>, dim3((unsigned)grid), dim3(CF_THREADS), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
