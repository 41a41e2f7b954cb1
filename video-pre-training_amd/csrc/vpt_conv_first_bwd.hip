// Backward of the fused stack-0 first convolution (weights and bias only: its input is the uint8 image).
//
// Forward (vpt_conv_first.hip): P = maxpool3x3/2( relu( conv3x3(img/255, W) + b ) ); the 128x128xC pre-pool tensor
// is never stored, so this kernel RECOMPUTES it tile by tile exactly as the forward does (same MFMA path, same
// bf16 rounding), then for every pooled pixel and channel finds the arg-max conv pixel of its 3x3 window (first
// maximum in scan order, torch's rule; a zero maximum passes no gradient through the ReLU), i.e. the sparse gradient
// G[conv pixel][channel] of the pre-pool tensor, and contracts it with the image patches on the matrix cores:
//     dW[o][kh][kw][ch] = sum_p G[p][o] * img[p + (kh,kw)][ch] / 255 ,   db[o] = sum_p G[p][o]
// is a GEMM with M = channels, N = 27 taps (+ one column of ones for db), K = the tile's 17 x 17 conv pixels.
//   1. recompute the conv tile into LDS (bf16, [289 pixels][128 channels]);
//   2. thread = (channel, half of the tile's 64 pooled pixels): arg-max search, positions kept in registers;
//   3. the conv tile is dead now: zero it and scatter the pooled gradients into it (ds_pk_add_bf16; a conv pixel can win up
//      to four overlapping windows, so G sums up to four bf16 values in bf16);
//   4. wave w contracts pixel slices w, w + 4, ... for all four 32-channel blocks: A = G^T through the LDS transpose read,
//      B = the patch matrix built from the tile's input bytes (taps x 16 pixels per slice); 4 x 16 fp32 accumulators per lane
//      persist over the workgroup's tiles and are reduced across the waves and flushed with atomics once at the end.
// Round 2: the first version did step 4 on the vector ALU -- 27 byte reads, 27 conversions and 27 FMAs per pooled pixel and
// thread, 3.5 ms per 1024 frames (8 TF/s) against 0.8 ms for the forward; VALU issue slots, not latency, were the limit
// (profiles/r02_ubench_mfma_valu.md).
// Persistent workgroups (2 per CU) sweep the tile list.  Replaces the autograd of lib/impala_cnn.py:86-97,115-117 for stack 0.
#include "vpt_common.h"
#include "vpt_kernels.h"

#ifndef VPT_CFB_ABLATE
#define VPT_CFB_ABLATE 0   // profiling builds: 1 no search, 2 no zero fill, 4 no scatter, 8 no MFMA contraction, 16 no recompute
#endif
#include "vpt_conv_first_tile.h"

__global__ __launch_bounds__(256, 2) void vpt_conv_first_bwd_kernel(VptConvFirstBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CF_SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: scalar (slice / pixel arithmetic of step 4 stays off the vector ALU)
  const int hi = lane >> 5, l31 = lane & 31;
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int tilesX = PW >> 3, tilesY = PH >> 3;
  const long T = (long)a.frames * tilesY * tilesX;
  const int nt = blockIdx.y;
  const int CB_out = a.Cout >> 5;

  op16x8 wfr[4][2];
#pragma unroll
  for (int cs = 0; cs < 4; ++cs)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wfr[cs][ks] = *((const op16x8*)a.wfrag + ((nt * 4 + cs) * 2 + ks) * 64 + lane);

  const int oc = tid & 127, half = tid >> 7;   // backward role: output channel within the N tile, pooled-pixel half
  const int og = nt * 128 + oc;
  const bool ovalid = og < a.Cout;
  f32x16 gacc[4];            // dW^T partial sums: [32-channel block][16 values]: rows = channels, column l31 = tap (27 = bias)
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[ob][r] = 0.f;
  // LDS transpose-read lane map (as in vpt_conv_wgrad.hip): 16-lane group g reads pixels 8 (g >> 1) + (i >> 2) (+ 4), channels 16 (g & 1) + 4 (i & 3)
  const int g16 = lane >> 4, i16 = lane & 15;
  const int tr_off = (8 * (g16 >> 1) + (i16 >> 2)) * CT_RS + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
  const int tap = l31;                                   // B operand row of this lane
  const int tap_c = min(tap, 26);
  const int tap_off = ((tap_c / 9) * 19 + (tap_c % 9) / 3) * 8 + tap_c % 3;   // byte of tap (kh, kw, ch) relative to the pixel's record: record (kh, kw), byte ch (taps >= 27: any valid byte)
  const unsigned char* in8 = smem + IN_OFF;

  // next tile's input bytes are fetched into registers while the current tile computes (as in the forward kernel)
  u32x2 nxt[CF_FETCH(256)];
  auto fetch = [&](int f, int ty, int tx) {
    cf_fetch_input<256>(a.img + (size_t)f * a.H * a.W * 3, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2, tid, nxt);
  };
  const long per = (T + gridDim.x - 1) / gridDim.x;
  const long t_begin = blockIdx.x * per, t_end = min(t_begin + per, T);
  // tile coordinates are decoded once and then counted up (no 64-bit divisions per tile)
  int tx, ty, f;
  {
    long L = t_begin;
    tx = (int)(L % tilesX); L /= tilesX;
    ty = (int)(L % tilesY);
    f = (int)(L / tilesY);
  }
  int ntx = tx, nty = ty, nf = f;
  if (t_begin < t_end) fetch(f, ty, tx);

  for (long tile = t_begin; tile < t_end; ++tile, tx = ntx, ty = nty, f = nf) {
    const int py0 = ty * 8, px0 = tx * 8;
    if (++ntx == tilesX) { ntx = 0; if (++nty == tilesY) { nty = 0; ++nf; } }
    cf_stage_input<256>(smem, nxt, tid);
    if (tid == 0) *(int*)(smem + CTR_OFF) = 0;
    __syncthreads();
    if (tile + 1 < t_end) fetch(nf, nty, ntx);
    // this thread's 32 pooled gradients, two bf16 per register, fetched in four groups of 8: group 0 now (its latency
    // hides behind the recompute), group g + 1 while group g is routed
    const unsigned short* dP = (const unsigned short*)a.dpooled + ((size_t)(f * CB_out + ((ovalid ? og : 0) >> 5)) * PH * PW) * 32 + ((ovalid ? og : 0) & 31);
    auto load_group = [&](int g, uint32_t* dst) {
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        const int pp = half * 32 + g * 8 + q;     // pp and pp + 1 are neighbours in the same pooled row
        const size_t o = (size_t)((py0 + (pp >> 3)) * PW + px0 + (pp & 7)) * 32;
        dst[q >> 1] = (uint32_t)dP[o] | ((uint32_t)dP[o + 32] << 16);
      }
    };
    uint32_t dreg[4][4];     // this thread's 32 pooled gradients, two bf16 per register (requested after the search)
    // ---- 1. recompute the conv tile (vpt_conv_first_tile.h: the forward kernel's code) ----
    if (!(VPT_CFB_ABLATE & 16)) cf_conv_tile(smem, wfr, lane, py0, px0, ty == 0 || tx == 0);
    __syncthreads();
    // ---- 2. arg-max search: conv pixel (0..288) of every pooled pixel, 0xffff = no gradient (ReLU gate / zero gradient) ----
    uint32_t cpk[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cpk[k] = 0xffffffffu;
    if (ovalid && !(VPT_CFB_ABLATE & 1)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
          const int pp = half * 32 + g * 8 + q8;
          const int pyl = pp >> 3, pxl = pp & 7;
          const short* ct = (const short*)(smem + ((2 * pyl) * 17 + 2 * pxl) * CT_RS) + oc;
          short best = 0;                                        // raw bf16 patterns as signed integers: only values > 0 can win (ReLU gate)
          int bpos = 0;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const short v = ct[(dy * 17 + dx) * (CT_RS / 2)];
              if (v > best) { best = v; bpos = dy * 17 + dx; }   // strict >: first maximum in scan order
            }
          const bool hit = best != 0;                            // ReLU gate (and windows whose maximum is 0)
          const uint32_t cpos = hit ? (uint32_t)((2 * pyl) * 17 + 2 * pxl + bpos) : 0xffffu;
          const int slot = g * 4 + (q8 >> 1);
          cpk[slot] = (q8 & 1) ? ((cpk[slot] & 0x0000ffffu) | (cpos << 16)) : ((cpk[slot] & 0xffff0000u) | cpos);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    // ---- 3. the conv tile becomes G: zero, then scatter ----
    if (ovalid) { load_group(0, dreg[0]); load_group(1, dreg[1]); load_group(2, dreg[2]); load_group(3, dreg[3]); }   // arrive under the zero fill
    if (!(VPT_CFB_ABLATE & 2))
    for (int i = tid; i < CT_BYTES / 16; i += 256) *(u32x4*)(smem + i * 16) = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
    if (VPT_CFB_ABLATE & 32) {   // profiling: keep the search alive without the scatter
      uint32_t x_ = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) x_ ^= cpk[k] + dreg[k >> 2][k & 3];
      if (x_ == 0x12345u) a.db[0] = 1.f;
    }
    if (ovalid && !(VPT_CFB_ABLATE & (4 | 32))) {
      // A conv pixel can win up to four overlapping windows (a 2 x 2 block of pooled pixels: index distances 1, 7, 8, 9), so G
      // sums up to four gradients.  They are merged in registers first -- the LAST pooled pixel of a group carries the sum --
      // and every (conv pixel, channel) entry is then WRITTEN once, no read-modify-write (32 LDS atomics per thread cost 2 ms
      // per 1024 frames, 32 dependent read-add-write round trips 1.3 ms).  Only conv row 8 is shared with the other half's
      // thread of this channel (windows of pooled rows 3 and 4): those entries are added with ds_pk_add_bf16.
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      float gs[32];
      uint32_t cp[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        cp[k] = (k & 1) ? (cpk[k >> 1] >> 16) : (cpk[k >> 1] & 0xffffu);
        const uint32_t dbits = (k & 1) ? (dreg[k >> 3][(k >> 1) & 3] >> 16) : (dreg[k >> 3][(k >> 1) & 3] & 0xffffu);
        gs[k] = op16_lo_to_f32(dbits);
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) {
#pragma unroll
        for (int dj = 0; dj < 4; ++dj) {
          const int off = (dj == 0) ? 1 : (6 + dj);            // 1, 7, 8, 9
          const int j = k - off;
          // j must be a real neighbour: same pooled row for off 1; previous row and column +1 / 0 / -1 for 7 / 8 / 9
          const bool nb = j >= 0 && ((off == 1) ? ((k & 7) != 0) : (off == 7) ? ((k & 7) != 7) : (off == 9) ? ((k & 7) != 0) : true);
          if (!nb) continue;
          const bool same = cp[j] == cp[k] && cp[k] != 0xffffu;
          gs[k] += same ? gs[j] : 0.f;
          cp[j] = same ? 0xffffu : cp[j];
        }
      }
      unsigned char* gcol = smem + oc * 2;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (cp[k] == 0xffffu) continue;
        const uint32_t vb = pack_op16x2(gs[k], 0.f) & 0xffffu;
        if (cp[k] - 8u * 17u < 17u) {          // shared row
          const uint32_t pair = (oc & 1) ? (vb << 16) : vb;
#ifdef VPT_OPERAND_F16
          typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
          __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h16x2*)(smem + (oc & ~1) * 2 + cp[k] * CT_RS), __builtin_bit_cast(h16x2, pair));
#else
          __builtin_amdgcn_ds_atomic_fadd_v2bf16((__attribute__((address_space(3))) s16x2*)(smem + (oc & ~1) * 2 + cp[k] * CT_RS), __builtin_bit_cast(s16x2, pair));
#endif
        } else {
          *(unsigned short*)(gcol + cp[k] * CT_RS) = (unsigned short)vb;
        }
      }
    }
    __syncthreads();
    // ---- 4. dW^T += G^T x patches on the matrix cores: wave w takes the 16-pixel slices w, w + 4, ... ----
    if (!(VPT_CFB_ABLATE & 8))
    for (int ks = w; ks < 19; ks += 4) {
      // B fragment: row = tap, k = conv pixels 16 ks + 8 hi .. + 7.  ks is wave-uniform, so the input-tile offset of every
      // (pixel, hi) pair is scalar arithmetic; per element one select (hi), one byte read of the pixel's record (the lane's tap
      // offset is in the base pointer) and one conversion.  Pixels beyond 288 need no masking here: the A side is exactly zero for them and bytes are
      // finite; taps 28..31 produce columns that are never flushed; tap 27 is the column of ones (db).
      u32x4 pk;
      uint32_t pw[4];
      const unsigned char* inl = in8 + tap_off;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float v2[2];
#pragma unroll
        for (int e1 = 0; e1 < 2; ++e1) {
          const int c0 = min(ks * 16 + e2 * 2 + e1, 288), c1 = min(ks * 16 + 8 + e2 * 2 + e1, 288);
          const int o0 = ((c0 / 17) * 19 + (c0 % 17)) * 8, o1 = ((c1 / 17) * 19 + (c1 % 17)) * 8;
          v2[e1] = (float)inl[hi ? o1 : o0];
        }
        pw[e2] = pack_op16x2_exact(v2[0], v2[1]);
      }
      const uint32_t ones = pack_op16x2_exact(1.0f, 1.0f);
      pk.x = (tap == 27) ? ones : pw[0]; pk.y = (tap == 27) ? ones : pw[1]; pk.z = (tap == 27) ? ones : pw[2]; pk.w = (tap == 27) ? ones : pw[3];
      const op16x8 bfrag = __builtin_bit_cast(op16x8, pk);
      // the last slice holds one real pixel (288): the seven rows behind it lie outside the tile -> masked to exact zeros
      const bool tail = ks == 18;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        const unsigned char* gp = smem + (ks * 16) * CT_RS + ob * 64 + tr_off;
        op16x4 a0 = lds_tr16_read(gp), a1 = lds_tr16_read(gp + 4 * CT_RS);
        if (tail) {
          u32x2 m0 = __builtin_bit_cast(u32x2, a0);
          m0.x = (hi == 0) ? (m0.x & 0xffffu) : 0u; m0.y = 0u;
          a0 = __builtin_bit_cast(op16x4, m0);
          a1 = __builtin_bit_cast(op16x4, (u32x2){0u, 0u});
        }
        const op16x8 afrag = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        gacc[ob] = VPT_MFMA_32X32X16(afrag, bfrag, gacc[ob], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- flush: reduce the four waves' partial sums through LDS, one atomic per (channel, tap) and workgroup ----
  float* red = (float*)smem;                               // [wave][128 channels][32 taps] fp32 = 64 KB
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      red[(w * 128 + ch) * 32 + l31] = gacc[ob][r];
    }
  __syncthreads();
  for (int i = tid; i < 128 * 28; i += 256) {
    const int ch = i / 28, k = i - ch * 28;
    const int o = nt * 128 + ch;
    if (o >= a.Cout) continue;
    const float v = (red[(0 * 128 + ch) * 32 + k] + red[(1 * 128 + ch) * 32 + k]) + (red[(2 * 128 + ch) * 32 + k] + red[(3 * 128 + ch) * 32 + k]);
    if (k < 27) atomicAdd(a.dw + (size_t)o * 27 + k, v * (1.0f / 255.0f));   // d(conv)/dW = img / 255
    else atomicAdd(a.db + o, v);
  }
}

extern "C" int vpt_conv_first_bwd_launch(const VptConvFirstBwdArgs* a, hipStream_t stream) {
  if ((a->H & 15) || (a->W & 15) || (a->Cout & 31) || a->frames <= 0) return -1;
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                 ? prop.multiProcessorCount : 256;
  }
  const long tiles = (long)a->frames * (a->H >> 4) * (a->W >> 4);
  if ((long)a->frames * a->H * a->W * 3 > 0x7fffffffL) return -2;   // 32-bit pixel offsets inside a launch
  long gx = (long)num_cu * 2;
  if (tiles < gx) gx = tiles;
  hipLaunchKernelGGL(vpt_conv_first_bwd_kernel, dim3((unsigned)gx, (a->Cout + 127) / 128), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
