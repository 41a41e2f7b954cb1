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
//   2. arg-max search, item = (pooled pixel, channel octet): nine 16-byte reads, per channel a signed key (pattern << 16 | 15 - scan index) and one
//      v_max_i32 per window position; the winning WINDOW OFFSET k* = 3 dy + dx (0..8; 15 = no gradient) as a 4-bit code;
//   3. the conv tile is dead now.  Into its LDS go (a) the table [64 pooled pixels][128 channels] of words (pooled gradient | k* << 16) -- the
//      gradients arrive by one 16-byte global load per item -- and (b) the B operand of step 4: patch values as 16-bit MFMA fragments, identical for
//      the waves of a pixel group, built cooperatively (36 k-steps x 64 lanes x 16 bytes);
//   4. NO scatter (round 4).  Write the sum over pooled pixels as nine sums over the window offsets:
//          dW[o][t] = sum_k sum_pp ( d[pp][o] * [k*(pp, o) == k] ) * X[2 pp + k][t]
//      -- for a FIXED offset the pooled pixels' conv pixels are distinct, so nothing collides and there is no gradient matrix G to zero, merge
//      and scatter into.  As a GEMM per wave: M = 32 channels (wave = channel block x pooled rows 0-3 / 4-7), N = 27 taps (+ a column of ones for
//      db), K = 32 pooled pixels x 9 offsets = 18 k-steps of 16.  K-slot order: k-step q < 16 = the 8 offsets k = 0..7 of pooled pixel q of the
//      wave's first row pair (slots 0..7, supplied by lanes hi = 0) and of its second (slots 8..15, hi = 1) -- the A fragment of a lane (channel,
//      row pair) is its OWN pixel's gradient placed at slot k*: a one-hot built in registers from the table word; k-steps 16, 17 = offset 8 of the
//      pixels 8 s .. 8 s + 7 of each row pair.  Twice the MFMAs of the scatter form (the matrix pipe is idle anyway); every gradient enters the
//      fp32 accumulation un-merged (the scatter form summed up to four bf16 values into one bf16 entry of G first).
//      16 fp32 accumulators per lane persist over the workgroup's tiles; the two waves of a channel block are added and written to the workgroup's row of a partial slab at the end
//      (summed in row order by vpt_slab_sum: bit-reproducible).
// History.  Round 2: step 4 on the vector ALU (27 byte reads, conversions and FMAs per pooled value): 3.5 ms per 1024 frames.  Round 3: G by
// merge-and-scatter + one MFMA contraction over the 289 conv pixels, thread = (channel, 32 pooled pixels), four waves: 2.0 ms; its ablation table
// charged 1.33 ms to the scatter -- but removing the scatter had let the compiler delete the search as dead code too.  Round 4 ablations on this
// kernel (profiles/r04_experiments.md section 13): the SEARCH was the cost (72 two-byte LDS reads + compare / select chains per thread at two
// waves per SIMD); wide reads + keys, eight waves, no scatter, gradients and codes through one LDS table: 1.24 ms.
// Persistent workgroups (2 per CU) sweep the tile list.  Replaces the autograd of lib/impala_cnn.py:86-97,115-117 for stack 0.
#include "vpt_common.h"
#include "vpt_kernels.h"

#ifndef VPT_CFB_ABLATE
#define VPT_CFB_ABLATE 0   // profiling builds: 1 no search, 8 no B build / MFMA contraction, 16 no recompute, 64 search reads one window position only
#endif
#include "vpt_conv_first_tile.h"

// Eight waves per workgroup, two workgroups per CU = four waves per SIMD (round 4; four waves per workgroup before): every phase of a tile is a chain
// of LDS round trips (search: 9 two-byte reads per pooled value; B build: byte reads; one-hot A + fragment read + MFMA) with a barrier behind it, and
// two waves per SIMD left the LDS latency exposed -- the same finding as for the forward kernel in round 3.  Needs <= 128 registers: 16 pooled pixels
// per thread instead of 32.
#define CFB_THREADS 512
// NFOLD (round 6): `dpooled` holds G = d loss / d n(P), the gradient BEHIND the stack's GroupNorm `n` (lib/impala_cnn.py:118-119), and the `n` backward is
// applied per element here -- d(P) = r (G gain - A - xhat B), xhat = (P - mu_P) r, (A, B) = pool_ab[f] / count from vpt_affine_bwd_reduce_kernel, exactly
// vpt_affine_bwd_apply_kernel's arithmetic and 16-bit rounding point.  P needs no load: the window maximum the arg-max search finds IS the pooled value
// (the recompute reproduces the forward bit for bit).  The separate apply pass over stack 0 (read P, read G, write dP: 3 MB per frame) disappears.
template <bool NFOLD>
__global__ __launch_bounds__(CFB_THREADS, 4) void vpt_conv_first_bwd_kernel(VptConvFirstBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CF_SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: scalar (slice / pixel arithmetic of step 4 stays off the vector ALU)
  const int hi = lane >> 5, l31 = lane & 31;
  const int PH = a.H >> 1, PW = a.W >> 1;
  const int tilesX = PW >> 3, tilesY = PH >> 3;
  const long T = (long)a.frames * tilesY * tilesX;
  const int nt = blockIdx.y;
  __shared__ int nt_stash_, bx_stash_;         // (read back by the flush behind the tile loop; the first barrier of the loop -- or the flush's own -- publishes them)
  if (tid == 0) { nt_stash_ = nt; bx_stash_ = blockIdx.x; }
  const int CB_out = a.Cout >> 5;

  const int cbw = w & 3, ph = w >> 2;          // wave = (32-channel block, pooled rows 0-3 / 4-7)
  const int oc = cbw * 32 + l31, quarter = ph * 2 + hi;   // backward role: output channel within the N tile, two pooled rows (16 pooled pixels)


  f32x16 gacc;               // dW^T partial sums of this wave's channel block: rows = channels, column l31 = tap (27 = bias)
#pragma unroll
  for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
  // B-fragment build role: fragment (global k-step w + 8 i, lane) for i = 0..4 -- this lane's tap and half, the wave's k-steps
  const int tap = l31;
  const int tap_c = min(tap, 26);
  const int tap_off = ((tap_c / 9) * 19 + (tap_c % 9) / 3) * 8 + tap_c % 3;   // byte of tap (kh, kw, ch) relative to the pixel's record: record (kh, kw), byte ch (taps >= 27: any valid byte)
  const unsigned char* inl = smem + IN_OFF + tap_off + hi * (4 * 19 * 8);    // lanes hi = 1: the pixel group's second pair of pooled rows = conv rows + 4

  // next tile's input bytes are fetched into registers while the current tile computes (as in the forward kernel)
  u32x2 nxt[CF_FETCH(CFB_THREADS)];
  auto fetch = [&](int f, int ty, int tx) {
    int tid_o = tid;                           // opaque per call: the record coordinates (tid / 19, tid % 19) are five instructions to recompute and two
    asm volatile("" : "+v"(tid_o));            // registers to keep across the tile loop -- the two that spilled in the NFOLD instantiation
    cf_fetch_input_assembled<CFB_THREADS>(a.img + (size_t)f * a.H * a.W * 3, a.H, a.W, 2 * (ty * 8) - 2, 2 * (tx * 8) - 2, tid_o, nxt);
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
    cf_stage_input_plain<CFB_THREADS>(smem + IN_OFF, nxt, tid);
    if (tid == 0) *(int*)(smem + CTR_OFF) = 0;
    __syncthreads();
    if (tile + 1 < t_end) fetch(nf, nty, ntx);
    // the pooled gradients of this thread's two search items (pooled pixel, channel octet): one 16-byte load each, requested now, used after the search
    u32x4 dv[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + CFB_THREADS * it;
      const int oct4 = item & 3, pxl = (0x76452310u >> (4 * ((item >> 2) & 7))) & 7, pyl = (item >> 5) & 7, cbl = item >> 8;
      const int og0 = nt * 128 + cbl * 32 + oct4 * 8;
      dv[it] = (u32x4){0u, 0u, 0u, 0u};
      if (og0 < a.Cout)
        dv[it] = *(const u32x4*)(a.dpooled + ((size_t)(f * CB_out + (og0 >> 5)) * PH * PW + (size_t)((py0 + pyl) * PW + px0 + pxl)) * 32 + (og0 & 31));
    }
    // ---- 1. recompute the conv tile (vpt_conv_first_tile.h: the forward kernel's code).  The weight fragments are fetched per tile (8 KB, L2):
    // resident for the whole workgroup they cost 32 of the 128 registers in every other phase. ----
    if (!(VPT_CFB_ABLATE & 16)) {
      op16x8 wfr[4][2];
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));      // opaque per tile: keeps the (loop-invariant) loads inside the loop
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wfr[cs][ks] = *((const op16x8*)a.wfrag + ((nt * 4 + cs) * 2 + ks) * 64 + lane_t);
      cf_conv_tile(smem, wfr, lane, py0, px0, ty == 0 || tx == 0);
    }
    __syncthreads();
    // ---- 2. arg-max search: window offset 3 dy + dx of every (pooled pixel, channel), 15 = no gradient (ReLU gate / zero maximum).
    // item = (pooled pixel, channel octet): nine ds_read_b128 (the 16 lanes of a read group take one conv pixel's 256 contiguous bytes) instead of
    // 72 two-byte reads -- the search was LDS-instruction-bound (round 4 ablation: 1.0 of the kernel's 2.0 ms per 1024 frames; the round-3 table
    // had charged that time to the scatter, whose removal had let the compiler delete the search as dead code).  First maximum in scan order
    // without compare / select chains: a signed 32-bit key = (16-bit pattern << 16) | (15 - scan index) per channel, one v_lshl_or / v_and_or and
    // one v_max_i32 per value: larger pattern wins, equal patterns keep the EARLIER position; a key below 0x10000 is a maximum <= 0. ----
    uint32_t kcode[2][2];    // [item][dword]: eight 4-bit codes (channel c of the octet at bits 4 c)
    float mp = 0.f, rpool = 1.f, nA = 0.f, nB = 0.f;
    if constexpr (NFOLD) {   // the frame's scalars of the `n` backward (broadcast loads; first used behind the first item's nine LDS reads)
      frame_mean_rstd(a.pool_stats, f, a.inv_count_pool, mp, rpool);
      nA = (float)(a.pool_ab[2 * f] * a.inv_count_pool);
      nB = (float)(a.pool_ab[2 * f + 1] * a.inv_count_pool);
      // wave-uniform: held in scalar registers across the two items (four vector registers were two too many for four waves per SIMD)
      mp = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mp)));
      rpool = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rpool)));
      nA = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, nA)));
      nB = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, nB)));
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + CFB_THREADS * it;
      // the forward kernel's pooling map (vpt_conv_first.hip): the four lane quads of a ds_read_b128 service group take pooled columns 0, 2, 4, 6 or
      // 1, 3, 5, 7 -- conflict-free; 16 contiguous lanes on one pixel's 256 bytes collide two by two (+0.3 ms per 1024 frames, measured)
      const int oct4 = item & 3, pxl = (0x76452310u >> (4 * ((item >> 2) & 7))) & 7, pyl = (item >> 5) & 7, cbl = item >> 8;
      unsigned src_o = (unsigned)(((2 * pyl) * 17 + 2 * pxl) * CT_RS + (cbl * 32 + oct4 * 8) * 2);
      asm volatile("" : "+v"(src_o));             // (opaque per tile, as for the B build below)
      const unsigned char* src = smem + src_o;
      int klo[4] = {0, 0, 0, 0}, khi[4] = {0, 0, 0, 0};
      f32x4 gl0 = {1.f, 1.f, 1.f, 1.f}, gl1 = gl0;
      if constexpr (NFOLD) {   // the octet's eight gains: requested before the LDS reads of the search, used after them (128 floats: L1 / L2 hits)
        const int og0 = nt * 128 + cbl * 32 + oct4 * 8;
        if (og0 < a.Cout) { gl0 = *(const f32x4*)(a.n_gain + og0); gl1 = *(const f32x4*)(a.n_gain + og0 + 4); }
      }
      if (!(VPT_CFB_ABLATE & 1)) {
#pragma unroll
        for (int dy = 0; dy < ((VPT_CFB_ABLATE & 64) ? 1 : 3); ++dy)
#pragma unroll
          for (int dx = 0; dx < ((VPT_CFB_ABLATE & 64) ? 1 : 3); ++dx) {
            const u32x4 v = *(const u32x4*)(src + (dy * 17 + dx) * CT_RS);
            const uint32_t c = 15u - (uint32_t)(dy * 3 + dx);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              klo[j] = max(klo[j], (int)((v[j] << 16) | c));
              khi[j] = max(khi[j], (int)((v[j] & 0xffff0000u) | c));
            }
          }
      }
      uint32_t codes = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t c0 = (klo[j] >= 0x10000) ? (15u - ((uint32_t)klo[j] & 15u)) : 15u;
        const uint32_t c1 = (khi[j] >= 0x10000) ? (15u - ((uint32_t)khi[j] & 15u)) : 15u;
        codes |= (c0 | (c1 << 4)) << (8 * j);
      }
      kcode[it][0] = codes; kcode[it][1] = 0u;
      if constexpr (NFOLD) {
        // G -> d(P) in place.  The pooled value of (pooled pixel, channel) is the maximum the keys carry (pattern = key >> 16; 0 where no
        // position is positive: the forward's pool starts from 0), so P is never loaded.
        float df[8];
        unpack8(dv[it], df);
        const float ng[8] = {gl0.x, gl0.y, gl0.z, gl0.w, gl1.x, gl1.y, gl1.z, gl1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t pw = ((uint32_t)klo[j] >> 16) | ((uint32_t)khi[j] & 0xffff0000u);
          const float p0 = op16_lo_to_f32(pw), p1 = op16_hi_to_f32(pw);
          const float xh0 = (p0 - mp) * rpool, xh1 = (p1 - mp) * rpool;
          df[2 * j] = rpool * (df[2 * j] * ng[2 * j] - nA - xh0 * nB);
          df[2 * j + 1] = rpool * (df[2 * j + 1] * ng[2 * j + 1] - nA - xh1 * nB);
        }
        dv[it] = pack8(df);          // (rounded to 16 bits where the separate pass stored it)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();         // every search read of the conv tile is done: its LDS becomes the B operand and the code table
    // (gradient, code) words -> LDS [64 pooled pixels][128 channels] (32 KB behind the 36 KB of B fragments): word = 16-bit gradient | code << 16,
    // two 16-byte writes per item; read back channel-major by the (channel, two pooled rows) threads of step 4, which replaces their 16 two-byte
    // global loads and 16 byte reads per thread.
#define CFB_WORD_OFF 40960
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + CFB_THREADS * it;
      const int oct4 = item & 3, pxl = (0x76452310u >> (4 * ((item >> 2) & 7))) & 7, pyl = (item >> 5) & 7, cbl = item >> 8;
      const uint32_t n = kcode[it][0];       // eight nibbles, channel c at bits 4 c
      u32x4 w0, w1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lo = (dv[it][j] & 0xffffu) | (((n >> (8 * j)) & 15u) << 16);
        const uint32_t hi2 = (dv[it][j] >> 16) | (((n >> (8 * j + 4)) & 15u) << 16);
        if (j < 2) { w0[2 * j] = lo; w0[2 * j + 1] = hi2; } else { w1[2 * (j - 2)] = lo; w1[2 * (j - 2) + 1] = hi2; }
      }
      unsigned char* dst = smem + CFB_WORD_OFF + ((pyl * 8 + pxl) * 128 + cbl * 32 + oct4 * 8) * 4;
      *(u32x4*)dst = w0;
      *(u32x4*)(dst + 16) = w1;
    }
    // ---- 3. B fragments: [pixel group ph][k-step 0..17][lane][16 bytes]; lane (tap, hi) of k-step q < 16 holds X[conv pixel of (pooled pixel
    // (2 ph + hi) * 16 + q, offset e)][tap], e = 0..7; of k-step 16 + s the offset-8 values of the pooled pixels (2 ph + hi) * 16 + 8 s + j,
    // j = 0..7.  One byte read + one conversion per element: the lane's tap and hi are in its base pointer, the k-step is wave-uniform, the element
    // an immediate offset. ----
    if (!(VPT_CFB_ABLATE & 8)) {
      unsigned inl_t = (unsigned)(size_t)(inl - smem);   // opaque per tile: left alone the compiler hoists ~50 per-element LDS addresses out of the tile loop (spills)
      asm volatile("" : "+v"(inl_t));
      int w_t = w;                                       // ... and the wave's k-step constants (five sets of scalar offsets) are re-derived per tile too: hoisted
      asm volatile("" : "+s"(w_t));                      // out of the tile loop they were the scalar registers that spilled (a dozen scalar instructions per tile)
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int ksg = w_t + 8 * i;                   // wave-uniform
        if (ksg < 36) {
          const int pg = ksg >= 18 ? 1 : 0, ks = ksg - 18 * pg;
          const unsigned char* pgb = smem + inl_t + pg * (8 * 19 * 8);                          // pixel group 1: pooled rows 4..7 = conv rows 8..
          float v[8];
          if (ks < 16) {
            const unsigned char* p0 = pgb + ((2 * (ks >> 3)) * 19 + 2 * (ks & 7)) * 8;     // pooled pixel (row ks >> 3 of the pair, column ks & 7)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)p0[((e / 3) * 19 + (e % 3)) * 8];
          } else {
            const unsigned char* p0 = pgb + ((2 * (ks - 16) + 2) * 19 + 2) * 8;            // offset (2, 2) of pooled row ks - 16 of the pair, columns 0..7
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)p0[(2 * e) * 8];
          }
          const uint32_t ones = pack_op16x2_exact(1.0f, 1.0f);
          u32x4 pk;
          pk.x = (tap == 27) ? ones : pack_op16x2_exact(v[0], v[1]);
          pk.y = (tap == 27) ? ones : pack_op16x2_exact(v[2], v[3]);
          pk.z = (tap == 27) ? ones : pack_op16x2_exact(v[4], v[5]);
          pk.w = (tap == 27) ? ones : pack_op16x2_exact(v[6], v[7]);
          *(u32x4*)(smem + (ksg * 64 + lane) * 16) = pk;
        }
        __builtin_amdgcn_sched_barrier(0);   // eight byte reads in flight
      }
    }
    __syncthreads();
    // ---- 4. dW^T += sum over the nine window offsets: A fragments from registers (one-hot placement of the lane's own gradient) ----
    if (!(VPT_CFB_ABLATE & 8)) {
      const unsigned char* bfr = smem + (ph * 18 * 64 + lane) * 16;
      uint32_t wd[16];                         // this thread's 16 (gradient | code << 16) words: channel oc, pooled pixels quarter * 16 + q
      {
        unsigned co = (unsigned)(CFB_WORD_OFF + ((quarter * 16) * 128 + oc) * 4);
        asm volatile("" : "+v"(co));
#pragma unroll
        for (int q = 0; q < 16; ++q) wd[q] = *(const uint32_t*)(smem + co + q * 512);
      }
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        u32x4 af;
        if (ks < 16) {
          const uint32_t d16 = wd[ks] & 0xffffu, code = wd[ks] >> 16;
          const uint32_t t = d16 << ((code & 1u) << 4);
          const uint32_t hs = code >> 1;                 // dword of the slot; 4 (offset 8) and 7 (no gradient) match none
          af.x = (hs == 0u) ? t : 0u; af.y = (hs == 1u) ? t : 0u; af.z = (hs == 2u) ? t : 0u; af.w = (hs == 3u) ? t : 0u;
        } else {
          const int sg = ks - 16;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {               // pixels 8 sg + 2 jj, + 1: offset 8
            const uint32_t a0 = wd[8 * sg + 2 * jj], a1 = wd[8 * sg + 2 * jj + 1];
            af[jj] = (((a0 >> 16) == 8u) ? (a0 & 0xffffu) : 0u) | (((a1 >> 16) == 8u) ? (a1 << 16) : 0u);
          }
        }
        const op16x8 bfrag = *(const op16x8*)(bfr + ks * 1024);
        gacc = VPT_MFMA_32X32X16(__builtin_bit_cast(op16x8, af), bfrag, gacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // one k-step at a time (left alone the scheduler hoists every fragment: > 128 registers)
      }
    }
    __syncthreads();
  }
  // ---- flush: the two waves of a channel block (pixel groups 0 / 1) are added through LDS, then the sums go to the workgroup's slab row ----
  // What the flush needs is derived HERE, behind the tile loop: the lane / wave roles from the thread index again (through a register the compiler
  // cannot see behind), the slab pointer and Cout from the kernarg segment, the N tile from an LDS word written at entry.  Taken from the values
  // of the kernel's first lines they had to stay live across the loop: 4 SGPRs + 1 VGPR spilled (8 bytes of scratch) until round 5.
  float* red = (float*)smem;                               // [4 channel blocks][16 values][64 lanes] fp32 = 16 KB
  int tid_f = threadIdx.x;
  asm volatile("" : "+v"(tid_f));
  const int lane_f = tid_f & 63, w_f = __builtin_amdgcn_readfirstlane(tid_f >> 6);
  const int hi_f = lane_f >> 5, l31_f = lane_f & 31, cbw_f = w_f & 3, ph_f = w_f >> 2;
  const __attribute__((address_space(4))) VptConvFirstBwdArgs* late = (const __attribute__((address_space(4))) VptConvFirstBwdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(late));
  const int cout_f = late->Cout;
  if (ph_f == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(cbw_f * 16 + r) * 64 + lane_f] = gacc[r];
  }
  __syncthreads();
  const int nt_f = __builtin_amdgcn_readfirstlane(nt_stash_);      // (behind a barrier in every path, also when the workgroup had no tile)
  float* prow = late->partials + (size_t)__builtin_amdgcn_readfirstlane(bx_stash_) * cout_f * 28;
  if (ph_f == 0 && l31_f < 28) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = nt_f * 128 + cbw_f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi_f;
      if (o >= cout_f) continue;
      const float v = gacc[r] + red[(cbw_f * 16 + r) * 64 + lane_f];
      // this workgroup's row of the partial slab, [dW Cout x 27 | db Cout]; the launcher's vpt_slab_sum adds the rows in row order (the tile ranges
      // are a static function of the grid, so a row's content does not depend on scheduling either).  Until round 5: one fp32 atomic per entry.
      if (l31_f < 27) prow[(size_t)o * 27 + l31_f] = v * (1.0f / 255.0f);   // d(conv)/dW = img / 255
      else prow[(size_t)cout_f * 27 + o] = v;
    }
  }
}

static long conv_first_bwd_grid_x(int frames, int H, int W) {
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                 ? prop.multiProcessorCount : 256;
  }
  const long tiles = (long)frames * (H >> 4) * (W >> 4);
  const long gx = (long)num_cu * 2;
  return tiles < gx ? tiles : gx;
}

// floats of the `partials` workspace: one row [Cout x 27 | Cout] per workgroup column + vpt_slab_sum's scratch
extern "C" long vpt_conv_first_bwd_partial_floats(int frames, int H, int W, int Cout) {
  const long gx = conv_first_bwd_grid_x(frames, H, W);
  return gx * Cout * 28 + vpt_slab_sum_scratch_floats((int)gx, Cout * 28);
}

extern "C" int vpt_conv_first_bwd_launch(const VptConvFirstBwdArgs* a, hipStream_t stream) {
  if ((a->H & 15) || (a->W & 15) || (a->Cout & 31) || a->frames <= 0 || !a->partials) return -1;
  if (a->n_gain && (!a->pool_stats || !a->pool_ab)) return -1;
  if ((long)a->frames * a->H * a->W * 3 > 0x7fffffffL) return -2;   // 32-bit pixel offsets inside a launch
  const long gx = conv_first_bwd_grid_x(a->frames, a->H, a->W);
  const dim3 grid((unsigned)gx, (a->Cout + 127) / 128);
  if (a->n_gain) hipLaunchKernelGGL(vpt_conv_first_bwd_kernel<true>, grid, dim3(CFB_THREADS), 0, stream, *a);
  else hipLaunchKernelGGL(vpt_conv_first_bwd_kernel<false>, grid, dim3(CFB_THREADS), 0, stream, *a);
  if (hipGetLastError() != hipSuccess) return -3;
  return vpt_slab_sum_launch(a->partials, (int)gx, a->Cout * 28, (long)a->Cout * 28, a->dw, a->Cout * 27, a->db, 1, a->partials + gx * a->Cout * 28, stream);
}
