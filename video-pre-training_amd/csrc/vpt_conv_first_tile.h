// The 17 x 17-pixel conv tile of the stack-0 first convolution, shared by the forward kernel (vpt_conv_first.hip) and by the
// backward kernel's recompute (vpt_conv_first_bwd.hip): both must produce the SAME 16-bit values.
//
// LDS image of one tile:   [0, CT_BYTES)          conv tile  [289 conv pixels][128 channels] 16-bit, pixel pitch CT_RS
//                          [IN_OFF, +IN_BYTES)    input tile [19][19][3] as 16-bit operands (the bytes 0..255, exact), +5 pad
//                          [CTR_OFF, +16)         work counter of the tile's 32-pixel slices
// A conv pixel's K vector is three runs of nine consecutive input values (kernel row dy: (kw, ch) contiguous at element
// ((cr + dy) * 19 + cc) * 3).  The 32 K slots of the two MFMA k-steps are ORDERED FOR THAT (vpt_pack.hip / packing.py use
// the same table):   ks 0, lanes 0-31 : row 0 values 0..7        ks 0, lanes 32-63 : row 1 values 0..7
//                    ks 1, lanes 0-31 : row 2 values 0..7        ks 1, lanes 32-63 : row 0 / 1 / 2 value 8, bias hi, bias lo, 0, 0, 0
// so a lane's fragment is one (2-byte aligned) 16-byte LDS read -- round 2 built it from 16 byte reads, 16 conversions and 8
// packs per k-step, 122 vector instructions per 8 MFMAs, and the kernel was vector-issue bound.
#pragma once
#include "vpt_common.h"

#define CT_RS 272
#define CT_BYTES (289 * CT_RS)  // 78608
#define IN_OFF CT_BYTES
#define IN_ELEMS (19 * 57)      // 1083
#define IN_BYTES 2176
#define CTR_OFF (IN_OFF + IN_BYTES)
#define CF_SMEM_BYTES (CTR_OFF + 16)

#ifdef VPT_OPERAND_F16
#define CF_ONE_BITS 0x3c00u
#else
#define CF_ONE_BITS 0x3f80u
#endif

// a byte 0..255 as the 16-bit operand's bit pattern (exact in both formats)
__device__ __forceinline__ unsigned short cf_byte_bits(unsigned char v) { return (unsigned short)(pack_op16x2_exact((float)v, 0.f) & 0xffffu); }

// this thread's five bytes of the 19 x 19 x 3 input tile (fetched one tile ahead) -> 16-bit LDS image
__device__ __forceinline__ void cf_stage_input(unsigned char* smem, const unsigned char (&nxt)[5], int tid) {
  unsigned short* in16 = (unsigned short*)(smem + IN_OFF);
#pragma unroll
  for (int m = 0; m < 5; ++m)
    if (tid + 256 * m < IN_ELEMS + 5) in16[tid + 256 * m] = (tid + 256 * m < IN_ELEMS) ? cf_byte_bits(nxt[m]) : (unsigned short)0;
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
  const unsigned short* ib = (const unsigned short*)(smem + IN_OFF) + (cr * 19 + cc) * 3;
  op16x8 pf[2];
  u32x4 y, l;
  __builtin_memcpy(&pf[0], ib + (hi ? 57 : 0), 16);
  __builtin_memcpy(&y, ib + 114, 16);
  l.x = (uint32_t)ib[8] | ((uint32_t)ib[57 + 8] << 16);
  l.y = (uint32_t)ib[114 + 8] | (CF_ONE_BITS << 16);
  l.z = CF_ONE_BITS;
  l.w = 0u;
  pf[1] = __builtin_bit_cast(op16x8, hi ? l : y);
  f32x16 acc[4];
#pragma unroll
  for (int cs = 0; cs < 4; ++cs) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cs][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc[cs] = VPT_MFMA_32X32X16(wfr[cs][ks], pf[ks], acc[cs], 0, 0, 0);
  }
  if (!pv) return;
  unsigned char* dst = smem + p * CT_RS + hi * 8;
  if (edge) {
    const int gy = 2 * py0 - 1 + cr, gx = 2 * px0 - 1 + cc;
    const uint32_t keep = (gy >= 0 && gx >= 0) ? 0xffffffffu : 0u;   // H, W are multiples of 16: the tile never crosses the bottom / right border
#pragma unroll
    for (int cs = 0; cs < 4; ++cs)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk2 = {pack_op16x2(acc[cs][4 * g + 0], acc[cs][4 * g + 1]) & keep, pack_op16x2(acc[cs][4 * g + 2], acc[cs][4 * g + 3]) & keep};
        *(u32x2*)(dst + (cs * 32 + g * 8) * 2) = pk2;
      }
  } else {
#pragma unroll
    for (int cs = 0; cs < 4; ++cs)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk2 = {pack_op16x2(acc[cs][4 * g + 0], acc[cs][4 * g + 1]), pack_op16x2(acc[cs][4 * g + 2], acc[cs][4 * g + 3])};
        *(u32x2*)(dst + (cs * 32 + g * 8) * 2) = pk2;
      }
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
