// The 17 x 17-pixel conv tile of the stack-0 first convolution, shared by the forward kernel (vpt_conv_first.hip) and by the
// backward kernel's recompute (vpt_conv_first_bwd.hip): both must produce the SAME 16-bit values.
//
// LDS image of one tile:   [0, CT_BYTES)          conv tile  [289 conv pixels][128 channels] 16-bit, pixel pitch CT_RS
//                          [IN_OFF, +IN_BYTES)    input tile as 19 x 19 eight-byte RECORDS: record (y, x) = the 8 input bytes that
//                                                 start at pixel (y, x), channel 0 (pixels x, x + 1 and two channels of x + 2)
//                          [CTR_OFF, +16)         work counter of the tile's 32-pixel slices
// A conv pixel's K vector is three runs of nine consecutive input bytes (kernel row dy: (kw, ch) contiguous).  The 32 K slots of
// the two MFMA k-steps are ORDERED FOR THAT (vpt_pack.hip / packing.py use the same table):
//     ks 0, lanes 0-31 : row 0 values 0..7        ks 0, lanes 32-63 : row 1 values 0..7
//     ks 1, lanes 0-31 : row 2 values 0..7        ks 1, lanes 32-63 : row 0 / 1 / 2 value 8, bias hi, bias lo, 0, 0, 0
// so a lane's fragment is one ALIGNED 8-byte LDS read of a record + eight byte -> operand conversions (values 8: byte 2 of
// record (y, x + 2)).  History: round 2 built it from 16 byte reads, 16 conversions and 8 packs per k-step; a 16-bit tile in its
// natural [y][x][ch] order read with 2-byte-aligned ds_read_b128 cost 1170 cycles of SQ_LDS_UNALIGNED_STALL per tile (PMC:
// profiles/r03_experiments.md section 8) -- the LDS pipe, 61 % busy with 30 % of that in stalls, is what bounds this kernel.
#pragma once
#include "vpt_common.h"

#define CT_RS 272
#define CT_BYTES (289 * CT_RS)  // 78608
#define IN_OFF CT_BYTES
#define IN_RECS (19 * 19)       // 361
#define IN_BYTES (IN_RECS * 8)  // 2888
#define CTR_OFF (IN_OFF + IN_BYTES)
#define CF_SMEM_BYTES (CTR_OFF + 16)

#ifdef VPT_OPERAND_F16
#define CF_ONE_BITS 0x3c00u
#else
#define CF_ONE_BITS 0x3f80u
#endif

// four bytes of a dword -> four 16-bit operands (0..255 are exact in both formats): v_cvt_f32_ubyteN + one pack per pair
__device__ __forceinline__ u32x2 cf_bytes4(uint32_t w) {
  u32x2 r;
  r.x = pack_op16x2_exact((float)(w & 0xffu), (float)((w >> 8) & 0xffu));
  r.y = pack_op16x2_exact((float)((w >> 16) & 0xffu), (float)(w >> 24));
  return r;
}
__device__ __forceinline__ u32x4 cf_bytes8(u32x2 rec) {
  const u32x2 a = cf_bytes4(rec.x), b = cf_bytes4(rec.y);
  return (u32x4){a.x, a.y, b.x, b.y};
}

// Fetch (one tile ahead, into registers) and staging of the input records.  Thread t < 361 owns record (t / 19, t % 19) of the
// tile whose input window starts at image pixel (iy0, ix0) = (2 py0 - 2, 2 px0 - 2).  The fetch is ONE unaligned 8-byte global load per
// record, unconditional and branch-free: a record that hangs over the left / right image border (pixels x .. x + 2 not all inside the row: two records per row
// in a quarter of the tiles) is loaded from its own address all the same -- the bytes beside the row belong to the neighbouring row -- with the
// address clamped into the frame (only the first row's left and the last row's right records move), and the staging shifts the bytes back and
// zeroes those outside the row (the conv's zero padding).  Until round 6 those records were assembled from up to six byte loads, each in its own
// divergent branch with its own s_waitcnt: six dependent global round trips in front of a border tile's compute.
#define CF_FETCH(THREADS) ((IN_RECS + (THREADS) - 1) / (THREADS))   // 2 per thread at 256 threads, 1 at 512
struct CfRecFix { int lo, hi, d; bool live; };     // valid bytes [lo, hi) of the record; d = wanted offset - loaded offset; live: any byte inside the image
__device__ __forceinline__ CfRecFix cf_rec_fix(int H, int W, int gy, int gx, int t, int& oc) {
  CfRecFix r;
  r.live = t < IN_RECS && gy >= 0 && gy < H && gx > -3 && gx < W;
  const int o = (gy * W + gx) * 3;
  oc = min(max(o, 0), H * W * 3 - 8);
  r.d = o - oc;
  r.lo = max(0, -3 * gx);
  r.hi = min(8, 3 * (W - gx));
  return r;
}
template <int THREADS>
__device__ __forceinline__ void cf_fetch_input(const uint8_t* img, int H, int W, int iy0, int ix0, int tid, u32x2 (&nxt)[CF_FETCH(THREADS)]) {
#pragma unroll
  for (int m = 0; m < CF_FETCH(THREADS); ++m) {
    const int t = tid + THREADS * m;
    const int y = t / 19, x = t - y * 19;
    int oc;
    const CfRecFix fx = cf_rec_fix(H, W, iy0 + y, ix0 + x, t, oc);
    (void)fx;
    __builtin_memcpy(&nxt[m], img + oc, 8);      // unconditional (the clamped address is inside the frame for every thread): records outside the image are zeroed by the staging
  }
}
template <int THREADS>
__device__ __forceinline__ void cf_stage_input(unsigned char* raw, const u32x2 (&nxt)[CF_FETCH(THREADS)], int tid, int H, int W, int iy0, int ix0) {
#pragma unroll
  for (int m = 0; m < CF_FETCH(THREADS); ++m) {
    const int t = tid + THREADS * m;
    if (t < IN_RECS) {
      const int y = t / 19, x = t - y * 19;
      int oc;
      const CfRecFix fx = cf_rec_fix(H, W, iy0 + y, ix0 + x, t, oc);
      u32x2 v = nxt[m];
      if (!fx.live) v = (u32x2){0u, 0u};
      else if (fx.lo | (8 - fx.hi) | fx.d) {               // border records only
        uint64_t r = (uint64_t)v.x | ((uint64_t)v.y << 32);
        r = fx.d < 0 ? r << (8 * -fx.d) : r >> (8 * fx.d);  // loaded from a clamped address: move the bytes to where they belong
        const uint64_t below_hi = fx.hi >= 8 ? ~0ull : ((1ull << (8 * fx.hi)) - 1ull);
        r &= below_hi & ~((1ull << (8 * fx.lo)) - 1ull);
        v.x = (uint32_t)r; v.y = (uint32_t)(r >> 32);
      }
      *(u32x2*)(raw + t * 8) = v;
    }
  }
}

// The backward kernel's fetch / staging (the form both kernels had until round 6): border records are assembled in the FETCH from byte loads under
// divergent branches and staged with a plain store.  In that kernel the fetch is issued behind the tile's first barrier, where the other waves' recompute
// covers its waits, while the staging sits in front of that barrier: the branch-free pair above (fix-up in the staging) measured 2.5 % slower there
// (1.20 -> 1.23 ms per 1024 frames), the opposite of the forward kernel, whose staging is off the critical path.
template <int THREADS>
__device__ __forceinline__ void cf_fetch_input_assembled(const uint8_t* img, int H, int W, int iy0, int ix0, int tid, u32x2 (&nxt)[CF_FETCH(THREADS)]) {
#pragma unroll
  for (int m = 0; m < CF_FETCH(THREADS); ++m) {
    const int t = tid + THREADS * m;
    const int y = t / 19, x = t - y * 19;
    const int gy = iy0 + y, gx = ix0 + x;
    u32x2 v = {0u, 0u};
    if (t < IN_RECS && gy >= 0 && gy < H) {
      const uint8_t* src = img + (gy * W + gx) * 3;
      if (gx >= 0 && gx + 2 < W) {
        __builtin_memcpy(&v, src, 8);
      } else {
        uint64_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int q = gx + j / 3;
          if (q >= 0 && q < W) acc |= (uint64_t)src[j] << (8 * j);
        }
        v.x = (uint32_t)acc; v.y = (uint32_t)(acc >> 32);
      }
    }
    nxt[m] = v;
  }
}
template <int THREADS>
__device__ __forceinline__ void cf_stage_input_plain(unsigned char* raw, const u32x2 (&nxt)[CF_FETCH(THREADS)], int tid) {
#pragma unroll
  for (int m = 0; m < CF_FETCH(THREADS); ++m)
    if (tid + THREADS * m < IN_RECS) *(u32x2*)(raw + (tid + THREADS * m) * 8) = nxt[m];
}

// One 32-pixel slice `sub` of the tile: conv + bias (1/255 is folded into the weights), rounded to 16 bits and stored RAW into
// the conv tile -- the ReLU commutes with the max-pool and is applied once per pooled value.  Pixels outside the image
// (the pool's padding row / column of the tiles on the top / left border: `edge`) are stored as 0.
__device__ __forceinline__ void cf_conv_slice(unsigned char* smem, const op16x8 (&wfr)[4][2], int sub, int lane, int py0, int px0, bool edge) {
  const int hi = lane >> 5, l31 = lane & 31;
  const int p = sub * 32 + l31;
  const bool pv = p < 289;
  const int pc = pv ? p : 288;
  const int cr = pc / 17, cc = pc - cr * 17;
  const unsigned char* ib = smem + IN_OFF + (cr * 19 + cc) * 8;
  const u32x2 x = *(const u32x2*)(ib + (hi ? 19 * 8 : 0));
  const u32x2 y = *(const u32x2*)(ib + 2 * 19 * 8);
  const float n0 = (float)ib[2 * 8 + 2], n1 = (float)ib[(19 + 2) * 8 + 2], n2 = (float)ib[(2 * 19 + 2) * 8 + 2];
  u32x4 l;
  l.x = pack_op16x2_exact(n0, n1);
  l.y = pack_op16x2_exact(n2, 1.0f);
  l.z = CF_ONE_BITS;
  l.w = 0u;
  op16x8 pf[2];
  pf[0] = __builtin_bit_cast(op16x8, cf_bytes8(x));
  pf[1] = __builtin_bit_cast(op16x8, hi ? l : cf_bytes8(y));
  unsigned char* dst = smem + p * CT_RS + hi * 8;
  uint32_t keep = 0xffffffffu;
  if (edge) {   // H, W are multiples of 16: a tile never crosses the bottom / right border
    const int gy = 2 * py0 - 1 + cr, gx = 2 * px0 - 1 + cc;
    keep = (gy >= 0 && gx >= 0) ? 0xffffffffu : 0u;
  }
  // two 32-channel blocks at a time: 32 live accumulator registers (the forward kernel runs four waves per SIMD, 128 registers)
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    f32x16 acc[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c2][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc[c2] = VPT_MFMA_32X32X16(wfr[2 * ch + c2][ks], pf[ks], acc[c2], 0, 0, 0);
    }
    if (pv) {
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 pk2 = {pack_op16x2(acc[c2][4 * g + 0], acc[c2][4 * g + 1]), pack_op16x2(acc[c2][4 * g + 2], acc[c2][4 * g + 3])};
          if (edge) { pk2.x &= keep; pk2.y &= keep; }
          *(u32x2*)(dst + ((2 * ch + c2) * 32 + g * 8) * 2) = pk2;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The tile's ten slices are handed out by an LDS counter (reset by the caller before the barrier in front of this call): 289
// pixels are 9 slices + 1 pixel, a static split gives two waves three slices and two waves two, and with two workgroups per CU
// the two heavy waves of both land on the same SIMDs.
__device__ __forceinline__ void cf_conv_tile(unsigned char* smem, const op16x8 (&wfr)[4][2], int lane, int py0, int px0, bool edge) {
  int* ctr = (int*)(smem + CTR_OFF);
  for (;;) {
    int sub = 0;
    if (lane == 0) sub = atomicAdd(ctr, 1);
    sub = __builtin_amdgcn_readfirstlane(sub);
    if (sub >= 10) break;
    cf_conv_slice(smem, wfr, sub, lane, py0, px0, edge);
  }
}
