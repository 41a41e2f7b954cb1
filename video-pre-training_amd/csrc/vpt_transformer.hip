// Non-GEMM pieces of the recurrent transformer trunk and the heads (gfx950), all fp32 arithmetic:
//
//  vpt_layernorm_kernel : nn.LayerNorm over the last dim (pre_r_ln lib/util.py:169,195; the norm of
//                         FanInInitReLULayer lib/util.py:61-62,78-79; final_ln lib/policy.py:188,214), optional
//                         ReLU on the input (F.relu at lib/policy.py:211), fp32 and/or bf16 output (the bf16
//                         copy is the next GEMM's A operand).  One wavefront per row, two-pass variance.
//  vpt_attn_kernel      : attention() (lib/xf.py:18-71) for the "clipped_causal" MaskedAttention
//                         (lib/masked_attention.py:38-41,75-83): per (sequence, head, 32-query tile) the band of
//                         31+maxlen keys drawn from [KV memory ; this chunk], logits q.k/d_head (muP scale,
//                         lib/xf.py:59) + relative-position bias R.b_nd (lib/xf.py:265-271, lib/util.py:232-267)
//                         + visibility (state_mask & !first for memory keys), fp32 softmax, P.V.  The
//                         [B*h, t, t+maxlen] bias / logit / weight tensors of the reference are never formed.
//  vpt_kv_update_kernel : SelfAttentionLayer.update_state (lib/xf.py:366-391): next memory = last `maxlen`
//                         rows of [memory ; new K/V], kept un-split in fp32 as the reference's state.
//  vpt_logsoftmax_kernel: CategoricalActionHead.forward (lib/action_head.py:170-174): logits / temperature,
//                         fp32 log_softmax over one head's column range; wavefront reductions.
#include "vpt_common.h"
#include "vpt_kernels.h"

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vpt_layernorm_kernel(VptLayerNormArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const float* x = a.x + (size_t)row * a.D;
  const int n4 = a.D >> 2;
  float s = 0.f;
  for (int i = lane; i < n4; i += 64) {
    f32x4 v = *(const f32x4*)(x + 4 * i);
    if (a.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = wave_sum(s) / (float)a.D;
  float ss = 0.f;
  for (int i = lane; i < n4; i += 64) {
    f32x4 v = *(const f32x4*)(x + 4 * i);
    if (a.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)a.D + VPT_NORM_EPS);
  for (int i = lane; i < n4; i += 64) {
    f32x4 v = *(const f32x4*)(x + 4 * i);
    if (a.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    const f32x4 g = *(const f32x4*)(a.gain + 4 * i), b = *(const f32x4*)(a.bias + 4 * i);
    f32x4 y;
    y.x = fmaf((v.x - mean) * rstd, g.x, b.x);
    y.y = fmaf((v.y - mean) * rstd, g.y, b.y);
    y.z = fmaf((v.z - mean) * rstd, g.z, b.z);
    y.w = fmaf((v.w - mean) * rstd, g.w, b.w);
    if (a.out_f32) *(f32x4*)(a.out_f32 + (size_t)row * a.D + 4 * i) = y;
    if (a.out_bf16) {
      u32x2 p = {pack_op16x2(y.x, y.y), pack_op16x2(y.z, y.w)};
      *(u32x2*)(a.out_bf16 + (size_t)row * a.D + 4 * i) = p;
    }
  }
}

extern "C" int vpt_layernorm_launch(const VptLayerNormArgs* a, hipStream_t stream) {
  if (a->M <= 0 || (a->D & 3)) return -1;
  hipLaunchKernelGGL(vpt_layernorm_kernel, dim3((a->M + 3) >> 2), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
#define ATT_DH 128
#define ATT_QT 32
#define ATT_NK 160                      // keys staged per query tile (31 + maxlen, maxlen <= 129)
#define ATT_RS (ATT_DH + 4)             // padded fp32 row
#define ATT_SS 164                      // padded score row (16-byte aligned: the P operand is read as float4)
#define ATT_S_OFF 0                                       // scores / probabilities [32][ATT_SS]
#define ATT_R_OFF (ATT_S_OFF + ATT_QT * ATT_SS)
#define ATT_B_OFF (ATT_R_OFF + ATT_QT * 10)
#define ATT_SC_OFF (ATT_B_OFF + 10 * 129)
#define ATT_PT_OFF ((ATT_SC_OFF + ATT_QT + 3) & ~3)       // partial logits of key tile 4: [wave][16][64]
#define ATT_FLOATS (ATT_PT_OFF + 4 * 16 * 64)

__global__ __launch_bounds__(256) void vpt_attn_kernel(VptAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ss = sm + ATT_S_OFF;
  float* Rs = sm + ATT_R_OFF;
  float* Bs = sm + ATT_B_OFF;
  float* Sc = sm + ATT_SC_OFF;

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y / a.heads, h = blockIdx.y - b * a.heads;
  const int q0 = blockIdx.x * ATT_QT;
  const int maxlen = a.maxlen, t = a.t, hid = a.hid;
  const size_t tok0 = (size_t)b * t;
  // causal (clipped band): staged key kk is row q0+1+kk of [memory ; chunk].  mask "none" (IDM,
  // lib/masked_attention.py:139-141 with maxlen = 0): every query sees all t <= ATT_NK rows of the chunk.
  const int jbase = a.causal ? q0 + 1 : 0;

  // ---- stage the R rows and b_nd (Q, K and V never pass through LDS: every element is used by exactly one lane) ----
  for (int idx = tid; idx < ATT_QT * 10; idx += 256) {
    const int r = idx / 10, n = idx - r * 10;
    Rs[idx] = (a.causal && q0 + r < t) ? a.qkvr[(tok0 + q0 + r) * a.ld + 3 * hid + h * 10 + n] : 0.f;
  }
  for (int idx = tid; idx < 10 * maxlen; idx += 256) Bs[idx] = a.b_nd[idx];
  __syncthreads();

  // ---- logits on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): D[query][key] = sum_d Q[query][d] K[key][d] ----
  // Round 2: this phase and P V below ran on the vector ALU with every query thread re-reading the K / V rows from LDS (5
  // ds_read_b128 per 16 FMAs): 20 TF/s.  A lane supplies Q[l31][d] and K[key tile * 32 + l31][d] for d = 8 g + 4 hi + e,
  // e = 0..3 -- one float4 each per four MFMAs (the k order of an MFMA step is free as long as both operands agree).
  // Key tiles 0..3 belong to waves 0..3; tile 4 (keys 128..159) is split over the waves by d (four groups each) and summed
  // through LDS, so every wave issues 80 MFMAs.
  const int l31 = lane & 31, hi = lane >> 5;
  // row j of [memory ; chunk] -> pointer to its K (which = 1) / V (which = 2) row of this head, or null beyond the chunk
  auto kv_row = [&](int j, int which) -> const float* {
    if (j < maxlen) return (which == 1 ? a.kmem : a.vmem) + ((size_t)b * maxlen + j) * hid + h * ATT_DH;
    if (j - maxlen < t) return a.qkvr + (tok0 + j - maxlen) * a.ld + which * hid + h * ATT_DH;
    return nullptr;
  };
  {
    f32x16 acc, acc4;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc4[r] = 0.f; }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const float* qa = (q0 + l31 < t) ? a.qkvr + (tok0 + q0 + l31) * a.ld + h * ATT_DH + 4 * hi : nullptr;
    const float* kr = kv_row(jbase + w * 32 + l31, 1);
    const float* kr4 = kv_row(jbase + 128 + l31, 1);
    const float* kb = kr ? kr + 4 * hi : nullptr;
    const float* kb4 = kr4 ? kr4 + 4 * hi : nullptr;
#pragma unroll 8
    for (int g = 0; g < 16; ++g) {
      const f32x4 q4 = qa ? *(const f32x4*)(qa + 8 * g) : z4, k4 = kb ? *(const f32x4*)(kb + 8 * g) : z4;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, k4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, k4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, k4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, k4.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int g = 4 * w + g4;
      const f32x4 q4 = qa ? *(const f32x4*)(qa + 8 * g) : z4, k4 = kb4 ? *(const f32x4*)(kb4 + 8 * g) : z4;
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, k4.x, acc4, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, k4.y, acc4, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, k4.z, acc4, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, k4.w, acc4, 0, 0, 0);
    }
    float* Pt = sm + ATT_PT_OFF;
#pragma unroll
    for (int r = 0; r < 16; ++r) Pt[(w * 16 + r) * 64 + lane] = acc4[r];
    __syncthreads();
    // scale, relative-position bias, visibility -> Ss.  Lane = key column; accumulator register r = query (r & 3) + 8 (r >> 2) + 4 hi.
    auto finish = [&](int kk, int r, float dot) {
      const int qi = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool qvalid = (q0 + qi) < t;
      const int off = maxlen - 1 + qi - kk;  // 0 = the query itself, maxlen-1 = oldest key in the band
      const int j = jbase + kk;
      bool vis = a.causal ? (qvalid && off >= 0 && off < maxlen) : (qvalid && kk < t);
      if (vis && a.causal && j < maxlen) vis = a.memvalid[(size_t)b * maxlen + j] != 0;
      float sc = -3.0e38f;
      if (vis) {
        float rb = 0.f;
        if (a.causal) {
#pragma unroll
          for (int n = 0; n < 10; ++n) rb = fmaf(Rs[qi * 10 + n], Bs[n * maxlen + off], rb);
        }
        sc = dot * (1.0f / ATT_DH) + rb;
      }
      Ss[qi * ATT_SS + kk] = sc;
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) finish(w * 32 + l31, r, acc[r]);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {   // key tile 4: this wave finishes accumulator registers 4 w .. 4 w + 3
      const int r = 4 * w + r4;
      const float dot = (Pt[(0 * 16 + r) * 64 + lane] + Pt[(1 * 16 + r) * 64 + lane]) + (Pt[(2 * 16 + r) * 64 + lane] + Pt[(3 * 16 + r) * 64 + lane]);
      finish(128 + l31, r, dot);
    }
  }
  __syncthreads();   // a softmax row collects the key tiles of all four waves
  // ---- softmax rows (8 per wave) ----
  for (int r = w * 8; r < w * 8 + 8; ++r) {
    float* srow = Ss + r * ATT_SS;
    const float s0 = srow[lane], s1 = srow[lane + 64], s2 = (lane + 128 < ATT_NK) ? srow[lane + 128] : -3.0e38f;
    const float m = wave_max(fmaxf(s0, fmaxf(s1, s2)));
    const float e0 = (s0 > -1.0e38f) ? expf(s0 - m) : 0.f;
    const float e1 = (s1 > -1.0e38f) ? expf(s1 - m) : 0.f;
    const float e2 = (s2 > -1.0e38f) ? expf(s2 - m) : 0.f;
    const float tot = wave_sum(e0 + e1 + e2);
    srow[lane] = e0;
    srow[lane + 64] = e1;
    if (lane + 128 < ATT_NK) srow[lane + 128] = e2;
    if (lane == 0) Sc[r] = (tot > 0.f) ? 1.0f / tot : 0.f;
  }
  __syncthreads();

  // ---- out = P V on the fp32 matrix cores: wave w owns d_head slice 32 w .. 32 w + 31; D[query][d] = sum_key P[query][key] V[key][d] ----
  {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const float* pa = Ss + l31 * ATT_SS + 4 * hi;             // A: row = query, k = keys 8 g + 4 hi + e (contiguous in the score row)
    const int dcol = w * 32 + l31;                             // B: column = d; its four keys' rows are wave-uniform per half-wave
#pragma unroll 4
    for (int g = 0; g < ATT_NK / 8; ++g) {
      const f32x4 p4 = *(const f32x4*)(pa + 8 * g);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* r0 = kv_row(jbase + 8 * g + e, 2);        // scalar arithmetic: the key index is uniform
        const float* r1 = kv_row(jbase + 8 * g + 4 + e, 2);
        const float* rr = hi ? r1 : r0;
        v[e] = rr ? rr[dcol] : 0.f;
      }
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.x, v[0], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.y, v[1], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.z, v[2], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.w, v[3], o, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (q0 + qi < t) a.out[(tok0 + q0 + qi) * hid + h * ATT_DH + w * 32 + l31] = (vpt_op16)(o[r] * Sc[qi]);
    }
  }
}

extern "C" int vpt_attn_launch(const VptAttnArgs* a, hipStream_t stream) {
  if (a->hid != a->heads * ATT_DH) return -1;
  if (a->causal ? (a->maxlen < 1 || a->maxlen > 129) : (a->maxlen != 0 || a->t > ATT_NK)) return -1;
  static unsigned long long optin_done = 0;
  const size_t lds = ATT_FLOATS * sizeof(float);
  if (!vpt_lds_optin((const void*)vpt_attn_kernel, (int)lds, &optin_done)) return -4;
  dim3 grid((a->t + ATT_QT - 1) / ATT_QT, a->B * a->heads);
  hipLaunchKernelGGL(vpt_attn_kernel, grid, dim3(256), lds, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// Acting step (t = 1, agent.py:190-206): one query per (sequence, head) against the maxlen - 1 newest memory rows and itself,
// FUSED with the recurrent-state update -- the K / V rows this step reads are exactly the rows the shifted memory keeps, so they
// are stored (one row up) as they pass through, and the visibility it derives (state_mask & ~first, lib/xf.py:366-391) is the
// next state mask.  The general kernel above spends ~40 us per layer on a 32-query MFMA tile with one live row, plus
// vpt_kv_update_kernel and five torch kernels for the masks; here a workgroup = (sequence, head) issues all its loads at once:
//   thread (key, half of d_head): 16 float4 of its K row -> dot with q, pair-sum by shuffle -> logit (+ rel-pos bias, visibility);
//   block softmax; thread (d, half of the keys): 64 V values (coalesced over d) x probabilities -> output.
// Every row is loaded into registers before a barrier and stored after it, and a workgroup owns its head's 128 columns of the
// memory: kout / vout MAY ALIAS kmem / vmem (the captured acting graph updates its state in place).  Without `done` mask_out must
// NOT alias state_mask: every head's workgroup reads the old mask, head 0's writes the new one, and workgroups are not ordered.
// With `done` ([B] ints, zero before the first launch) the mask may be updated in place too: every workgroup counts itself in
// after its mask bytes are in registers, and the one that arrives LAST writes the new mask and resets the counter -- the captured
// acting graph then needs no mask copies (four memcpy nodes per step).
// Same formulas as vpt_attn_kernel (scale 1 / d_head, bias sum_n R[n] b_nd[n][off], invisible rows excluded, all-invisible -> 0);
// the sums run on the vector ALU in a different order than the matrix cores', so results agree to fp32 rounding, not bit for bit.
struct VptAttnStepExtra {
  const uint8_t* state_mask;   // [B][maxlen] bool bytes: the memory rows that hold data
  const uint8_t* first;        // [B] bool bytes: episode start -> the memory is ignored (and invalidated)
  uint8_t* mask_out;           // [B][maxlen]: next step's state_mask
  float* kout;                 // [B][maxlen][hid]
  float* vout;
  int* done;                   // optional [B]: arrival counter of the sequence's workgroups (in-place mask update)
};

__global__ __launch_bounds__(256) void vpt_attn_step_kernel(VptAttnArgs a, VptAttnStepExtra x) {
  __shared__ float sc_[128], red_[8], part_[128];
  __shared__ int last_;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
  const int maxlen = a.maxlen, hid = a.hid;
  const float* qrow = a.qkvr + (size_t)b * a.ld;            // t = 1: token b
  // row j of [memory ; new token], this head's slice: K (which = 1) or V (which = 2)
  auto kv_row = [&](int j, int which) -> const float* {
    if (j < maxlen) return (which == 1 ? a.kmem : a.vmem) + ((size_t)b * maxlen + j) * hid + h * ATT_DH;
    return qrow + which * hid + h * ATT_DH;
  };
  // ---- logits: thread = (key kk, half) ----
  const int kk = tid >> 1, half = tid & 1;
  const bool live = kk < maxlen;
  const int j = 1 + kk;
  f32x4 kv[16];
  float s = -3.0e38f;
  bool vis = false;
  if (live) {
    // every load of the phase first (K row, q, visibility, the rel-pos operands), then the arithmetic
    const float* kr = kv_row(j, 1) + half * 64;
    const float* q = qrow + h * ATT_DH + half * 64;
    f32x4 qv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kv[i] = *(const f32x4*)(kr + 4 * i);
#pragma unroll
    for (int i = 0; i < 16; ++i) qv[i] = *(const f32x4*)(q + 4 * i);
    const unsigned char mv = x.state_mask[(size_t)b * maxlen + min(j, maxlen - 1)];
    const unsigned char fst = x.first[b];
    const int off = maxlen - 1 - kk;
    float rr[10], bb[10];
#pragma unroll
    for (int n = 0; n < 10; ++n) { rr[n] = qrow[3 * hid + h * 10 + n]; bb[n] = a.b_nd[n * maxlen + off]; }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dot = fmaf(qv[i].x, kv[i].x, dot); dot = fmaf(qv[i].y, kv[i].y, dot); dot = fmaf(qv[i].z, kv[i].z, dot); dot = fmaf(qv[i].w, kv[i].w, dot);
    }
    dot += __shfl_xor(dot, 1, 64);
    vis = (j < maxlen) ? (mv != 0 && fst == 0) : true;     // memory row: state_mask & ~first; the token itself: always
    if (vis) {
      float rb = 0.f;
#pragma unroll
      for (int n = 0; n < 10; ++n) rb = fmaf(rr[n], bb[n], rb);
      s = dot * (1.0f / ATT_DH) + rb;
    }
  }
  // ---- block softmax over the keys (every key sits in two lanes: count half 0 only) ----
  float m = wave_max(s);
  if (lane == 0) red_[w] = m;
  __syncthreads();           // (all K rows and mask bytes of the workgroup are in registers now: the shifted stores may alias them)
  if (live) {
    float* ko = x.kout + ((size_t)b * maxlen + kk) * hid + h * ATT_DH + half * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) *(f32x4*)(ko + 4 * i) = kv[i];   // memory row kk of the next step = row kk + 1 of [memory ; new]
    if (!x.done && h == 0 && half == 0) x.mask_out[(size_t)b * maxlen + kk] = vis ? 1 : 0;   // = cat(state_mask[1:] & ~first, [True])
  }
  if (x.done && tid == 0) {     // this workgroup's mask reads are complete (they fed the barrier above): count in
    __threadfence();
    const int arrived = atomicAdd(x.done + b, 1);
    last_ = (arrived == a.heads - 1);
    if (last_) x.done[b] = 0;   // the next launch starts from zero again (kernel boundary orders it)
  }
  m = fmaxf(fmaxf(red_[0], red_[1]), fmaxf(red_[2], red_[3]));
  const float e = (s > -1.0e38f) ? expf(s - m) : 0.f;
  const float tot_w = wave_sum(half == 0 ? e : 0.f);
  if (half == 0) sc_[kk] = e;                              // (kk = tid >> 1 < 128)
  if (lane == 0) red_[4 + w] = tot_w;
  __syncthreads();
  const float tot = (red_[4] + red_[5]) + (red_[6] + red_[7]);
  const float inv = (tot > 0.f) ? 1.0f / tot : 0.f;
  if (x.done && last_ && live && half == 0) x.mask_out[(size_t)b * maxlen + kk] = vis ? 1 : 0;   // every workgroup of b has read the old mask
  // ---- out = P V: thread = (d, half of the keys); all 64 values in registers before the barrier, shifted stores after it ----
  {
    const int d = tid & 127, kh = tid >> 7;
    const int k0 = kh * 64;
    float v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = kv_row(1 + min(k0 + i, maxlen - 1), 2)[d];
    float o = 0.f;
    int cnt = __builtin_amdgcn_readfirstlane(maxlen - k0);       // rows k0 .. k0 + cnt - 1 of this half exist (wave-uniform)
#pragma unroll
    for (int i = 0; i < 64; ++i)
      if (i < cnt) o = fmaf(sc_[k0 + i], v[i], o);
    if (kh == 1) part_[d] = o;
    __syncthreads();
    asm volatile("" : "+s"(cnt));   // the 64 row tests are re-derived here: shared with the loop above they sat in 64 scalar registers across the barrier (54 spills)
#pragma unroll
    for (int i = 0; i < 64; ++i)
      if (i < cnt) x.vout[((size_t)b * maxlen + k0 + i) * hid + h * ATT_DH + d] = v[i];
    if (kh == 0) a.out[(size_t)b * hid + h * ATT_DH + d] = (vpt_op16)((o + part_[d]) * inv);
  }
}

extern "C" int vpt_attn_step_launch(const VptAttnArgs* a, const uint8_t* state_mask, const uint8_t* first, uint8_t* mask_out, float* kout, float* vout,
                                    int* done, hipStream_t stream) {
  if (a->hid != a->heads * ATT_DH || a->t != 1 || !a->causal || a->maxlen < 1 || a->maxlen > 128) return -1;
  if (!kout || !vout || !state_mask || !first || !mask_out || (mask_out == state_mask && !done)) return -1;
  VptAttnStepExtra x;
  x.state_mask = state_mask; x.first = first; x.mask_out = mask_out; x.kout = kout; x.vout = vout; x.done = done;
  hipLaunchKernelGGL(vpt_attn_step_kernel, dim3((unsigned)(a->B * a->heads)), dim3(256), 0, stream, *a, x);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vpt_kv_update_kernel(VptKvUpdateArgs a) {
  const int which = blockIdx.y;  // 0 = K, 1 = V
  const size_t n4 = (size_t)a.B * a.maxlen * (a.hid >> 2);
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n4) return;
  const int h4 = a.hid >> 2;
  const int c4 = (int)(idx % h4);
  const size_t rs = idx / h4;
  const int s = (int)(rs % a.maxlen);
  const int b = (int)(rs / a.maxlen);
  const int src = s + a.t;  // row of [memory ; new]
  const float* mem = which ? a.vmem : a.kmem;
  float* out = which ? a.vout : a.kout;
  f32x4 v;
  if (src < a.maxlen) v = *(const f32x4*)(mem + ((size_t)b * a.maxlen + src) * a.hid + c4 * 4);
  else v = *(const f32x4*)(a.qkvr + ((size_t)b * a.t + (src - a.maxlen)) * a.ld + (which ? 2 : 1) * a.hid + c4 * 4);
  *(f32x4*)(out + ((size_t)b * a.maxlen + s) * a.hid + c4 * 4) = v;
}

extern "C" int vpt_kv_update_launch(const VptKvUpdateArgs* a, hipStream_t stream) {
  if (a->B <= 0 || (a->hid & 3)) return -1;
  const size_t n4 = (size_t)a->B * a->maxlen * (a->hid >> 2);
  dim3 grid((unsigned)((n4 + 255) / 256), 2);
  hipLaunchKernelGGL(vpt_kv_update_kernel, grid, dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// NW waves per row.  The scaled logits of a thread (the first LSM_CAP it owns) stay in registers for the three passes -- one
// load and ONE division by the temperature per element (the reference divides: z / T, not z * (1 / T)).  Rows are independent:
// a batch of rows (M >= 64: training / chunked inference) takes 4 waves per row, the acting path's one or two rows of 8641
// classes take 16 (34 elements and three dependent passes per thread were 17-31 us per launch of a 1 ms step).
#define LSM_CAP 9
template <int NW>
__global__ __launch_bounds__(64 * NW) void vpt_logsoftmax_kernel(VptLogSoftmaxArgs a) {
  constexpr int NT = 64 * NW;
  __shared__ float red[3 * NW];
  __shared__ int redi[NW];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* z = a.logits + (size_t)row * a.ld + a.col0;
  const uint8_t* mk = a.mask ? a.mask + (size_t)row * a.n : nullptr;
  const float T = a.temperature;
#define SCALED(i_) ((mk && !mk[i_]) ? -100.0f : z[i_] / T)   /* shaped_out /= T; shaped_out[~mask] = LOG0 */
  float c[LSM_CAP];
  float m = -3.0e38f;
#pragma unroll
  for (int k = 0; k < LSM_CAP; ++k) {
    const int i = tid + NT * k;
    c[k] = (i < a.n) ? SCALED(i) : -3.0e38f;
    m = fmaxf(m, c[k]);
  }
  for (int i = tid + NT * LSM_CAP; i < a.n; i += NT) m = fmaxf(m, SCALED(i));
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) m = fmaxf(m, red[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LSM_CAP; ++k) s += (tid + NT * k < a.n) ? expf(c[k] - m) : 0.f;
  for (int i = tid + NT * LSM_CAP; i < a.n; i += NT) s += expf(SCALED(i) - m);
  s = wave_sum(s);
  if (lane == 0) red[NW + w] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < NW; ++k) tot += red[NW + k];
  const float lse = m + logf(tot);
  float* o = a.out + (size_t)row * a.n;
  // CategoricalActionHead.sample (lib/action_head.py:195-207) on the way out: argmax of the log-probs, or of
  // log-probs - log(-log u) (Gumbel-max; u == 1 -> 0.999 as the reference guards); FIRST maximum, as torch.argmax.
  // The uniforms are the caller's (a.noise) or generated here from device-resident state (a.rng_state: seed + step counter,
  // vpt_philox_uniform) -- the form a captured acting step needs: no buffer, no host-side generator, a new draw per replay.
  const float* u = a.noise ? a.noise + (size_t)row * a.n : nullptr;
  const bool gen = !u && a.rng_state != nullptr;
  uint64_t seed = 0, rstep = 0;
  if (gen) { seed = a.rng_state[0]; rstep = a.rng_state[1]; }
  float best = -3.0e38f;
  int besti = 0x7fffffff;
  auto emit = [&](int i, float sc_) __attribute__((always_inline)) {
    const float lp = sc_ - lse;
    o[i] = lp;
    if (a.action) {
      float sc = lp;
      if (u || gen) {
        float ui = u ? u[i] : vpt_philox_uniform(seed, rstep, a.rng_stream, (uint32_t)row, (uint32_t)i);
        if (ui == 1.0f) ui = 0.999f;
        sc = lp - logf(-logf(ui));
      }
      if (sc > best) { best = sc; besti = i; }     // ascending i per thread: keeps the first maximum
    }
  };
#pragma unroll
  for (int k = 0; k < LSM_CAP; ++k)
    if (tid + NT * k < a.n) emit(tid + NT * k, c[k]);
  for (int i = tid + NT * LSM_CAP; i < a.n; i += NT) emit(i, SCALED(i));
  if (a.action) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(besti, off, 64);
      if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) { red[2 * NW + w] = best; redi[w] = besti; }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < NW; ++k)
        if (red[2 * NW + k] > best || (red[2 * NW + k] == best && redi[k] < besti)) { best = red[2 * NW + k]; besti = redi[k]; }
      a.action[row] = besti;
      if (a.action_logp) a.action_logp[row] = ((mk && !mk[besti]) ? -100.0f : z[besti] / T) - lse;
    }
  }
#undef SCALED
}

extern "C" int vpt_logsoftmax_launch(const VptLogSoftmaxArgs* a, hipStream_t stream) {
  if (a->M <= 0 || a->n <= 0) return -1;
  if (a->M < 64 && a->n > 1024) hipLaunchKernelGGL((vpt_logsoftmax_kernel<16>), dim3(a->M), dim3(1024), 0, stream, *a);
  else hipLaunchKernelGGL((vpt_logsoftmax_kernel<4>), dim3(a->M), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------------------
// Tail of MinecraftAgentPolicy.act on the acting path (lib/policy.py:307-327), one launch instead of eight ATen kernels inside the
// captured step: log_prob = sum of the heads' action log-probs, vpred de-normalised (lib/normalize_ewma.py:27-31: v * std + mean),
// the NaN assertion as a flag, everything the caller keeps packed into ONE buffer (a single clone per step):
//   keep[b] = { buttons action, camera action, (float bits) log_prob | 0, (float bits) vpred de-normalised | raw vpred << 32 }
__global__ void vpt_act_epilogue_kernel(const int64_t* act_b, const int64_t* act_c, const float* lp_b, const float* lp_c, const float* logits, int ld,
                                        int vcol, float scale, float shift, int64_t* keep, uint8_t* nan_flag, uint64_t* rng_state, int B) {
  const int b = threadIdx.x;
  bool bad = false;
  if (b < B) {
    const float lp = lp_b[b] + lp_c[b];
    const float v = logits[(size_t)b * ld + vcol];
    const float vd = fmaf(v, scale, shift);
    bad = lp != lp;
    keep[4 * b + 0] = act_b[b];
    keep[4 * b + 1] = act_c[b];
    keep[4 * b + 2] = (int64_t)(uint64_t)__builtin_bit_cast(uint32_t, lp);
    keep[4 * b + 3] = (int64_t)((uint64_t)__builtin_bit_cast(uint32_t, vd) | ((uint64_t)__builtin_bit_cast(uint32_t, v) << 32));
  }
  const unsigned long long any = __ballot(bad);
  if (threadIdx.x == 0) {
    *nan_flag = any ? 1 : 0;
    if (rng_state) rng_state[1] += 1;    // the step's draws are done (both head launches precede this one in stream order): next step, next counter
  }
}

// The uniforms vpt_logsoftmax_kernel generates for (rng_state, rng_stream), written out [M][n] -- for callers that want to see / replay a
// draw (the tests feed them back through the `noise` argument: the two paths must pick the same actions).  Does NOT advance the state.
__global__ __launch_bounds__(256) void vpt_uniform_noise_kernel(const uint64_t* rng_state, uint32_t rng_stream, float* out, int M, int n) {
  const uint64_t seed = rng_state[0], rstep = rng_state[1];
  const long total = (long)M * n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int row = (int)(idx / n), i = (int)(idx - (long)row * n);
    out[idx] = vpt_philox_uniform(seed, rstep, rng_stream, (uint32_t)row, (uint32_t)i);
  }
}

extern "C" int vpt_uniform_noise_launch(const uint64_t* rng_state, uint32_t rng_stream, float* out, int M, int n, hipStream_t stream) {
  if (!rng_state || !out || M <= 0 || n <= 0) return -1;
  const long total = (long)M * n;
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(vpt_uniform_noise_kernel, dim3(grid), dim3(256), 0, stream, rng_state, rng_stream, out, M, n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int vpt_act_epilogue_launch(const int64_t* act_b, const int64_t* act_c, const float* lp_b, const float* lp_c, const float* logits, int ld,
                                       int vcol, float scale, float shift, int64_t* keep, uint8_t* nan_flag, uint64_t* rng_state, int B, hipStream_t stream) {
  if (B <= 0 || B > 64 || !keep || !nan_flag) return -1;
  hipLaunchKernelGGL(vpt_act_epilogue_kernel, dim3(1), dim3(64), 0, stream, act_b, act_c, lp_b, lp_c, logits, ld, vcol, scale, shift, keep, nan_flag, rng_state, B);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
