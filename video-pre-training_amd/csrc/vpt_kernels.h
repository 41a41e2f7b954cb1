// Internal kernel argument blocks + launcher prototypes (one launcher per .hip file).
// The public C ABI lives in include/vpt_hip.h and is implemented in vpt_capi.hip on top of these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef VPT_OPERAND_F16
typedef _Float16 vpt_op16;   // precision mode "fp16" (see vpt_common.h)
#else
typedef __bf16 vpt_op16;
#endif

struct VptConv3x3Args {
  const vpt_op16* x;       // [F][Cin/32][H][W][32]
  const vpt_op16* wpk;     // [NT][Cin/32][9][128][32], GroupNorm gain folded
  const float* edge_sa;    // [9][CoutPad]
  const float* edge_sg;    // [9][CoutPad]
  const double* stats_in;  // [F][2]  sum / sumsq of x
  const vpt_op16* res;     // optional residual, same layout as y
  vpt_op16* y;             // [F][Cout/32][H][W][32]
  double* stats_out;       // optional [F][2], accumulated (caller zeroes)
  int frames, H, W, Cin, Cout, CoutPad, NT;
  double inv_count_in;     // 1 / (Cin*H*W)
  int tiling;              // forward only: 1 = throughput kernel (16x16 px x 128 couts), 2 = latency kernel (x 32 couts)
  int ablate;              // profiling only (env VPT_CONV_ABLATE)
  long long* trace;        // profiling only: per-workgroup phase timestamps (vpt_conv3x3_set_trace)
  // GroupNorm `n` of the stack folded into its first block (forward, optional): kk_frame [F][9][CoutPad] = this frame's whole epilogue table and
  // rs_frame [F] = its accumulator scale (replace edge_sa / the statistics of x: vpt_nfold_coef_kernel); res_scale [F] + res_bias [F][Cout]:
  // the residual enters as res_scale[f] * res + res_bias[f][channel]
  const float* kk_frame;
  const float* rs_frame;
  const float* res_scale;
  const float* res_bias;
  // pool-fused forward (pool != 0, no residual): y is the POOLED output [F][Cout/32][H/2][W/2][32]; the tiles' last rows / columns go to
  // seam_r [F][Cout/32][H/16][W][32] / seam_c [F][Cout/32][W/16][H][32]; stats_out receives the pooled pixels that are complete in-tile
  int pool;
  vpt_op16* seam_r;
  vpt_op16* seam_c;
  vpt_op16* pool_mask;     // pool mode, optional (training forward, mode 7): per pooled value the 9-bit "window position differs from the maximum" mask, uint16
  double* chs_out;         // pool mode, optional [F][Cout][2], accumulated: per-channel (sum, sum of squares) of the stored complete pixels
  const float* out_gain;   // pool mode, optional [Cout]: the pooled pixels that are complete in-tile are stored multiplied by it (GroupNorm `n`'s gain,
                           // folded: the seam kernel does the same for the others); the statistics are those of the unscaled values
  // dgrad mode (bwd != 0): no GroupNorm fold, no ReLU; out = conv + res + coef[f][0] + coef[f][1] * xin
  int bwd;
  const vpt_op16* xin;     // the forward layer's input x (same shape as this call's output)
  const float* coef;       // [F][2]
  // gated dgrad (bwd != 0, no res; round 5): the output is the backward operand of the res-free layer whose output is xin --
  // out = rstd_g[f] * (conv + c0 + c1 * xin) * [xin > 0], rstd_g from gate_stats (that layer's input statistics); gate_u [F] += sum of
  // rstd_g * (conv + c0 + c1 * xin) * xin (accumulated; caller zeroes)
  const double* gate_stats;
  double inv_count_gate;
  double* gate_u;
};

struct VptChannelStatsArgs {   // per-frame, per-channel sums of a blocked tensor: chs[f][c] = (sum_p x, sum_p x^2), accumulated in fp64
  const vpt_op16* x;       // [F][CB][HW][32]
  double* chs;             // [F][CB*32][2]  (caller zeroes)
  int frames, CB, HW, split;   // split: workgroups per (frame, channel block), set by the launcher
};

struct VptNfoldCoefArgs {      // GroupNorm `n` of a stack folded into its first block: per-frame coefficients
  const double* tot;       // [F][2] sum / sum of squares of the pooled tensor P (all channels)
  const double* chs;       // [F][C][2] per-channel sums of Q = gain_n * P
  const float* gain;       // [C] n.weight
  const float* bias;       // [C] n.bias
  const float* sa;         // [9][CoutPad] edge tables of block 0's conv0: SA, SG (vpt_pack_conv3x3) and TB / TG = its rounded weights summed
  const float* sg;         //   against n.bias / n.weight instead of ones
  const float* tb;
  const float* tg;
  float* kk_frame;         // [F][9][CoutPad]  out: conv0's epilogue table of the frame
  float* rs_frame;         // [F]              out: conv0's accumulator scale r_x * r_P
  float* res_scale;        // [F]              out: conv1's residual = res_scale * Q + res_bias[c]
  float* res_bias;         // [F][C]
  int frames, C, HW, CoutPad;
};

struct VptPoolSeamArgs {    // finishes the pooled pixels whose 3x3 window crosses a tile border of the pool-fused convolution
  vpt_op16* y;             // pooled [F][CB][H/2][W/2][32]: the seam pixels hold the in-tile part of their maximum
  const vpt_op16* seam_r;  // [F][CB][H/16][W][32]  row 15 of every tile row
  const vpt_op16* seam_c;  // [F][CB][W/16][H][32]  column 15 of every tile column
  double* stats_out;       // [F][2] accumulated: the seam pixels' share of the pooled frame's statistics
  const float* gain;       // optional [CB*32]: the finished pixels are stored multiplied by it (statistics: of the unscaled values)
  double* chs_out;         // optional [F][CB*32][2], accumulated: per-channel (sum, sum of squares) of the finished pixels as stored
  int frames, CB, H, W;    // H, W: the PRE-pool size
  vpt_op16* mask;          // optional (training forward): the arg-max masks [F][CB][H/2][W/2][32] uint16 of vpt_conv3x3_kernel mode 7, finished here
};

struct VptPackConvArgs {
  const float* weight;     // [Cout][Cin][3][3]
  const float* gain;       // [Cin]  GroupNorm weight
  const float* bias;       // [Cin]  GroupNorm bias (unused when edge_sa is null)
  vpt_op16* wpk;           // [NT][Cin/32][9][128][32]
  float* edge_sa;          // optional [9][NT*128]
  float* edge_sg;          // optional [9][NT*128]
  int Cout, Cin, NT;
};

struct VptConvFirstArgs {
  const uint8_t* img;      // [F][H][W][3]
  const vpt_op16* wfrag;   // [NT][4][2][64][8]  MFMA A-operand fragments (bias folded in k=27,28)
  vpt_op16* y;             // pooled output [F][Cout/32][H/2][W/2][32]
  double* stats_out;       // [F][2]
  const float* out_gain;   // optional [Cout]: the pooled output is stored multiplied by it (statistics: of the unscaled values)
  double* chs_out;         // optional [F][Cout][2], accumulated: per-channel (sum, sum of squares) of the stored output (Cout <= 128)
  int frames, H, W, Cout, NT;
};

struct VptConv3dArgs {
  const uint8_t* img;      // [B*T][H][W][3]
  const vpt_op16* wfrag;   // [NT][4][64][8]  MFMA A-operand fragments, k = dt*3 + ch (15 used)
  const float* bias;       // [NT*128]
  vpt_op16* y;             // [B*T][Cout/32][H][W][32]
  double* stats_out;       // [B*T][2]
  int frames, T, H, W, Cout, NT;
};

struct VptPoolArgs {
  const vpt_op16* x;       // [F][CB][H][W][32]  (non-negative values: post-ReLU)
  vpt_op16* y;             // [F][CB][H/2][W/2][32]
  double* stats_out;       // [F][2]
  uint8_t* argmax;         // optional [F][CB][H/2][W/2][32]: window position code kh*3+kw of the first maximum (15: window all zero)
  int frames, CB, H, W;
};

struct VptAffineArgs {     // y = (x - mean_f) * rstd_f * g[idx] + b[idx]
  const vpt_op16* x;
  vpt_op16* y;
  const float* gain;       // per channel [C] (per_element=0) or per position [C*H*W] in blocked order (=1)
  const float* bias;
  const double* stats_in;  // [F][2]
  double* stats_out;       // optional [F][2]
  int frames, CB, HW, per_element;
  double inv_count;
};

struct VptGemmArgs {
  const vpt_op16* A;       // [M][lda] bf16 row-major
  const vpt_op16* wpk;     // [NT][K/32][128][32]
  const float* bias;       // [N] or null
  const float* res;        // [M][ldr] fp32 or null
  float* out_f32;          // [M][ldc] or null
  vpt_op16* out_bf16;      // [M][ldcb] or null
  int M, N, K, lda, ldr, ldc, ldcb;
  int relu, splitk, atomic_out;
  int tiling;              // 0: M <= 8 rows take the weight-streaming kernel (vpt_gemv.hip), more the MFMA GEMM; 1: MFMA GEMM whatever M; 2: weight-streaming (M <= 8); 4: MFMA GEMM, the 256 x 256 / eight-wave kernel where its grid fills the chip (bit-identical; A/B)
  const vpt_op16* mask;    // optional [M][ldm]: output is zeroed where mask <= 0 (ReLU backward)
  int ldm;
  // fused LayerNorm prologue (skinny path only, M <= 8, K <= 3072): A = op16(LayerNorm(ln_x)) computed by every workgroup
  const float* ln_x;       // [M][K] fp32 or null (then A is used)
  const float* ln_gain;    // [K]
  const float* ln_bias;    // [K]
  float* ln_out_f32;       // optional [M][K]: the normalised row in fp32 (residual / latent), written by workgroup 0
  int ln_relu_in;
};

struct VptGemmTnArgs {     // C[n1][n2] (+)= sum_m A[m][n1] * B[m][n2]
  const vpt_op16* A;       // [M][lda]
  const vpt_op16* B;       // [M][ldb]
  float* C;                // [N1][ldc]
  int M, N1, N2, lda, ldb, ldc, accumulate;
};

struct VptLayerNormArgs {
  const float* x;          // [M][D]
  const float* gain;
  const float* bias;
  float* out_f32;          // optional [M][D]
  vpt_op16* out_bf16;      // optional [M][D]
  int M, D, relu_in;
};

struct VptAttnArgs {
  const float* qkvr;       // [B*t][ld]: Q | K | V | R(10 per head, head-major)
  const float* kmem;       // [B][maxlen][hid]
  const float* vmem;
  const uint8_t* memvalid; // [B][maxlen]  state_mask & !first
  const float* b_nd;       // [10][maxlen]
  vpt_op16* out;           // [B*t][hid]
  int B, t, heads, hid, ld, maxlen, causal;
};

struct VptKvUpdateArgs {
  const float* qkvr;       // new K at col hid, V at col 2*hid
  const float* kmem;
  const float* vmem;
  float* kout;
  float* vout;
  int B, t, hid, ld, maxlen;
};

struct VptLogSoftmaxArgs {
  const float* logits;     // [M][ld]
  float* out;              // [M][n]
  int M, ld, col0, n;
  float temperature;
  const uint8_t* mask;     // optional [M][n]: 0 = action not available -> its scaled logit is LOG0 = -100 (lib/action_head.py:170-171)
  const float* noise;      // optional [M][n] uniforms in [0, 1]: Gumbel-max sampling (lib/action_head.py:198-207); null = argmax
  const uint64_t* rng_state;  // optional device {seed, step}: the uniforms are generated in the kernel instead (Philox4x32-10 keyed by the seed,
  uint32_t rng_stream;        //   counter = (element / 4, row, step, stream): vpt_philox_uniform in vpt_common.h); rng_stream tells one head's draw from another's
  long* action;            // optional [M]: sampled / arg-max index (first maximum)
  float* action_logp;      // optional [M]: out[row][action[row]]
};

struct VptAffineBwdArgs {
  const vpt_op16* x;       // the affine's INPUT, blocked
  const vpt_op16* dy;      // gradient w.r.t. its output
  const vpt_op16* dx_add;  // optional, added to dx (pass 2)
  vpt_op16* dx;            // pass 2 output
  const float* gain;
  const double* stats_in;  // statistics of x
  double* ab;              // [F][2] sum dy g, sum dy g xhat (pass 1 accumulates, pass 2 reads)
  float* dgain;            // accumulated
  float* dbias;
  float* partials;         // workspace of passes 1 (per-channel gain) and 3: vpt_affine_bwd_partial_floats
  int frames, CB, HW, per_element;
  double inv_count;
};

struct VptPoolBwdArgs {
  const vpt_op16* pre;     // pre-pool tensor [F][CB][H][W][32]
  const vpt_op16* pooled;  // [F][CB][H/2][W/2][32]
  const vpt_op16* dpooled;
  vpt_op16* dpre;
  int frames, CB, H, W;
};

struct VptConvBwdPrepArgs {
  const vpt_op16* dy;      // gradient w.r.t. the layer output (after ReLU and residual add); null -> (dpooled, argmax)
  const vpt_op16* dpooled; // gradient w.r.t. max_pool(y) [F][CB][H/2][W/2][32]   (fused max-pool backward)
  const uint8_t* argmax;   // window position code kh*3+kw of the maximum, same shape (vpt_pool_kernel)
  float* sbuf;             // scratch (vpt_conv_bwd_prep_scratch_floats): [F][9*Cout + Cout/32] fp32 per-frame edge-class sums of dz and per-plane sums dz v,
                           // then [ceil(F/32)][2][9*Cout] partial sums of dSA / dSG per block of 32 frames
  int wshift;              // log2(W), filled by the launcher
  const vpt_op16* y;       // saved layer output
  const vpt_op16* res;     // saved residual input or null
  const double* stats_in;  // statistics of the conv's INPUT x
  const float* edge_sa;    // [9][CoutPad]
  const float* edge_sg;
  vpt_op16* dacc;          // rstd * dz, blocked like y
  double* t12;             // optional [F][2]: T1 = sum dz (v - SA), T2 = sum dz SG   (written)
  float* coef;             // [F][2]: (c0, c1) of the input gradient's statistics terms dx += c0 + c1 x   (written)
  float* d_sa;             // [9][CoutPad] accumulated
  float* d_sg;
  int frames, CB, H, W, CoutPad;
  double inv_count_in;     // 1 / (Cin*H*W)
  // pre-gated variant (round 5): `dy` IS the operand dacc = rstd dz already (written by the gated dgrad, vpt_conv3x3_kernel mode 6); y, res and
  // dacc are unused, nothing is written but the per-frame sums: S[e][o] = (sum dacc) / rstd, and the data term sum dz v = gate_u[f] / rstd
  const double* gate_u;
  // pooled variant (round 5; dy, y, res, argmax unused): the layer feeds the max-pool and was run pool-fused with arg-max masks (vpt_conv3x3_kernel
  // mode 7): pooled P [F][CB][H/2][W/2][32] and pool_mask (same shape, uint16) replace the pre-pool tensor and the arg-max bytes
  const vpt_op16* pooled;
  const vpt_op16* pool_mask;
  // ... with the stack's GroupNorm `n` backward applied on the fly (n_gain != null): `dpooled` then holds G = the gradient w.r.t. x = n(P), and
  // d(pooled) = r_P (G gain - A - xhat B), xhat = (P - mu_P) r_P, (A, B) = pool_ab[f] / count from vpt_affine_bwd_reduce_kernel -- the affine
  // backward's second pass (read P, read G, write dP) disappears: this kernel reads P and a gradient tensor anyway
  const float* n_gain;     // [Cout]
  const double* pool_stats; // [F][2] sum / sum of squares of P
  const double* pool_ab;   // [F][2]
  double inv_count_pool;   // 1 / (Cout * H/2 * W/2)
};

struct VptConvFirstBwdArgs {
  const uint8_t* img;      // [F][H][W][3]
  const vpt_op16* wfrag;   // forward weight fragments (the pre-pool tile is recomputed)
  const vpt_op16* dpooled; // gradient w.r.t. the pooled output [F][Cout/32][H/2][W/2][32]
  float* dw;               // [Cout][27] in (kh, kw, ch) order, accumulated
  float* db;               // [Cout] accumulated
  float* partials;         // workspace: vpt_conv_first_bwd_partial_floats
  int frames, H, W, Cout;
  // optional (n_gain != null): `dpooled` is the gradient BEHIND the stack's GroupNorm `n`, whose backward is applied per element on the fly
  const float* n_gain;     // [Cout]
  const double* pool_stats; // [F][2] sum / sum of squares of the pooled tensor P
  const double* pool_ab;   // [F][2] (sum G gain, sum G gain xhat) from vpt_frame_affine_backward's pass 1
  double inv_count_pool;   // 1 / (Cout * H/2 * W/2)
};

struct VptConvWgradArgs {
  const vpt_op16* dacc;    // [F][Cout/32][H][W][32]
  const vpt_op16* x;       // [F][Cin/32][H][W][32]
  float* dw;               // [Cout][9][Cin] accumulated (caller zeroes)
  float* partial;          // scratch [vpt_conv_wgrad_groups()][Cout][9][Cin]
  int frames, H, W, Cin, Cout;
  int OT, frames_per_wg;   // set by the launcher
};

struct VptNllBwdArgs {
  const float* lp_buttons; // [M][nb] log-probabilities (forward output)
  const float* lp_camera;  // [M][nc]
  const long* act_buttons; // [M]
  const long* act_camera;  // [M]
  vpt_op16* dz;            // [M][ldz]: d loss / d (pre-softmax logits), columns nb+nc.. zero
  int M, nb, nc, ldz;
  float scale;             // 1 / (frames in the global batch * temperature)
};

struct VptHeadsBwdArgs {   // generic backward of the two log-softmax heads + the value column (autograd boundary)
  const float* lp_buttons; // [M][nb] log-probabilities (forward output)
  const float* lp_camera;  // [M][nc]
  const float* g_buttons;  // optional [M][nb]: incoming d loss / d log-prob
  const float* g_camera;   // optional [M][nc]
  const float* g_value;    // optional [M]: incoming d loss / d (raw value-head output)
  const uint8_t* mask_buttons;  // optional [M][nb] / [M][nc]: 0 = the logit was replaced by the constant LOG0 in the forward
  const uint8_t* mask_camera;   //   (lib/action_head.py:170-171): no gradient reaches it
  vpt_op16* dz;            // [M][ldz]: d loss / d (fused head logits), padding columns zero
  float grad_scale;        // multiplies every written value (fp16 loss scaling; 1 otherwise)
  int M, nb, nc, ldz;
  float inv_temp;          // 1 / temperature
};

struct VptGateCastArgs {
  const float* x;          // [M][ldx]
  const vpt_op16* mask;    // optional [M][ldm]
  vpt_op16* out;           // [M][ldo]
  int M, N, ldx, ldm, ldo;
};

struct VptLnBwdArgs {
  const float* x;          // [M][D] the LayerNorm's input (before the optional ReLU)
  const float* gain;       // [D]
  const float* dy;         // [M][D]
  const float* dx_add;     // optional [M][D] added to the result (skip connections)
  float* dx;               // [M][D]
  float* dgain;            // [D] accumulated (caller zeroes)
  float* dbias;            // [D]
  float* partials;         // [4 * ceil(M / 32)][2][D] workspace: per-wave column sums, added in row order by the finish kernel
  int M, D, relu_in;
#ifdef VPT_LN_DEBUG
  float* debug;            // diagnostics build only (tools/ubench/pk_hazard)
#endif
};

struct VptClipArgs {
  const uint8_t* src;          // [F][H][W][3] decoded frames, BGR
  const int* cursor;           // [F][3] (gui open, cursor x, cursor y) or null: no compositing
  const uint8_t* cursor_img;   // [CH][CW][3] BGR
  const double* cursor_alpha;  // [CH][CW] opacity in 0..1
  uint8_t* dst;                // [F][OH][OW][3] RGB
  int frames, H, W, OH, OW, CH, CW;
  double scale_x, scale_y;     // filled by the launcher
  int area2x2;                 // filled by the launcher
};

struct VptColsumArgs {
  const vpt_op16* x;       // [M][ld]
  float* out;              // [N] accumulated (caller zeroes)
  float* partials;         // [vpt_colsum_partial_floats(M, N)] workspace (row-slice sums, added in slice order)
  int M, N, ld;
};

struct VptAttnBwdArgs {
  const float* qkvr;       // forward projections [B*t][ld]
  const float* kmem;       // [B][maxlen][hid] (detached state: no gradient)
  const float* vmem;
  const uint8_t* memvalid; // [B][maxlen]
  const float* b_nd;       // [10][maxlen]
  const float* dout;       // [B*t][hid] gradient w.r.t. the merged attention output
  float* dqkvr;            // [B*t][ld]: every column written (Q, R by the main kernel, K, V by the finish kernel)
  float* db_nd;            // [10][maxlen] accumulated (caller zeroes)
  float* dkv_slab;         // [5][B*t][2 hid] workspace: per query tile pieces of dK / dV (vpt_attn_bwd_dkv_floats)
  float* dbnd_slab;        // workspace: one db_nd row per workgroup + the slab sum's scratch (vpt_attn_bwd_dbnd_floats)
  int B, t, heads, hid, ld, maxlen;
};

struct VptAdamTensor {      // one parameter tensor of a multi-tensor Adam step (device array, sorted by first_block)
  float* p;
  const float* g;
  float* m;
  float* v;
  unsigned long long n;     // elements
  long long first_block;    // first 1024-element block of the launch that belongs to this tensor
};

struct VptAdamArgs {
  float* p;                // parameters (updated in place)
  const float* g;          // gradients
  float* m;                // first moment
  float* v;                // second moment
  size_t n;
  float beta1, beta2, eps, weight_decay, grad_scale;
  float step_size;         // lr / (1 - beta1^t)
  float inv_sqrt_bc2;      // 1 / sqrt(1 - beta2^t)
  const int* skip_flag;    // optional device flag (multi-tensor form): non-zero = leave every tensor untouched (a loss-scaled step
                           // whose gradients overflowed, vpt_grads_nonfinite_multi)
};

extern "C" {
int vpt_pack_conv_first_launch(const float* w, const float* bias, void* out, int Cout, hipStream_t s);
int vpt_pack_conv3d_t5_launch(const float* w, const float* bias, void* out, float* bias_pad, int O, hipStream_t s);
int vpt_chw_to_blocked_launch(const float* src, float* dst, long rows, int C, int H, int W, hipStream_t s);
int vpt_adam_launch(const VptAdamArgs* a, hipStream_t s);
int vpt_adam_multi_launch(const VptAdamTensor* table_dev, int ntensors, long total_blocks, const VptAdamArgs* h, hipStream_t s);
int vpt_grads_nonfinite_launch(const VptAdamTensor* table_dev, int ntensors, long total_blocks, int* flag, hipStream_t s);
int vpt_affine_bwd_launch(const VptAffineBwdArgs* a, int pass, hipStream_t s);
long vpt_affine_bwd_partial_floats(int frames, int CB, int HW, int per_element, int pass);
int vpt_pool_bwd_launch(const VptPoolBwdArgs* a, hipStream_t s);
int vpt_conv_bwd_prep_launch(const VptConvBwdPrepArgs* a, hipStream_t s);
int vpt_conv_first_bwd_launch(const VptConvFirstBwdArgs* a, hipStream_t s);
long vpt_conv_first_bwd_partial_floats(int frames, int H, int W, int Cout);
long vpt_conv_bwd_prep_scratch_floats(int frames, int Cout);
int vpt_gemm_tn_launch(const VptGemmTnArgs* a, hipStream_t s);
int vpt_splitk_epilogue_launch(const float* part, int splitk, const VptGemmArgs* a, hipStream_t s);
int vpt_dense_fold_epilogue_launch(const float* part, int splitk, const double* stats, double inv_count, const float* sg, const float* sb, float* out, int M, int N, hipStream_t s);
int vpt_camera_codec_launch(int decode, const void* in, void* out, long n, double maxval, double binsize, double mu, int mu_law, hipStream_t s);
int vpt_action_mapping_launch(int to_factored, const long* a, const long* b, long* oa, long* ob, long n, int n_camera_bins, hipStream_t s);
int vpt_conv_wgrad_groups(int frames, int Cin, int Cout);
int vpt_conv_wgrad_launch(const VptConvWgradArgs* a, hipStream_t s);
int vpt_pack_conv3x3_launch(const VptPackConvArgs* a, hipStream_t s);
int vpt_pack_linear_launch(const float* w, void* out, int N, int K, int transposed, int ldw, int src_rows, hipStream_t s);
int vpt_nll_bwd_launch(const VptNllBwdArgs* a, hipStream_t s);
int vpt_heads_bwd_launch(const VptHeadsBwdArgs* a, hipStream_t s);
int vpt_ln_bwd_launch(const VptLnBwdArgs* a, hipStream_t s);
int vpt_gate_cast_launch(const VptGateCastArgs* a, hipStream_t s);
int vpt_colsum_launch(const VptColsumArgs* a, hipStream_t s);
int vpt_clip_launch(const VptClipArgs* a, hipStream_t s);
int vpt_attn_step_launch(const VptAttnArgs* a, const uint8_t* state_mask, const uint8_t* first, uint8_t* mask_out, float* kout, float* vout, int* done, hipStream_t s);
int vpt_act_epilogue_launch(const int64_t* act_b, const int64_t* act_c, const float* lp_b, const float* lp_c, const float* logits, int ld, int vcol,
                            float scale, float shift, int64_t* keep, uint8_t* nan_flag, uint64_t* rng_state, int B, hipStream_t s);
int vpt_uniform_noise_launch(const uint64_t* rng_state, uint32_t rng_stream, float* out, int M, int n, hipStream_t s);
int vpt_attn_bwd_launch(const VptAttnBwdArgs* a, hipStream_t s);
long vpt_attn_bwd_dkv_floats(int B, int t, int hid);
long vpt_attn_bwd_dbnd_floats(int B, int t, int heads, int maxlen);
long vpt_colsum_partial_floats(int M, int N);
long vpt_slab_sum_scratch_floats(int rows, int cols);
int vpt_slab_sum_launch(const float* slab, int rows, int cols, long ld, float* out_a, int split, float* out_b, int accumulate, float* scratch, hipStream_t s);
int vpt_conv3x3_launch(const VptConv3x3Args* a, hipStream_t s);
int vpt_pool_seam_launch(const VptPoolSeamArgs* a, hipStream_t s);
int vpt_channel_stats_launch(const VptChannelStatsArgs* a, hipStream_t s);
int vpt_nfold_coef_launch(const VptNfoldCoefArgs* a, hipStream_t s);
int vpt_conv_first_launch(const VptConvFirstArgs* a, hipStream_t s);
int vpt_conv3d_launch(const VptConv3dArgs* a, hipStream_t s);
int vpt_pool_launch(const VptPoolArgs* a, hipStream_t s);
int vpt_affine_launch(const VptAffineArgs* a, hipStream_t s);
int vpt_gemm_launch(const VptGemmArgs* a, hipStream_t s);
int vpt_layernorm_launch(const VptLayerNormArgs* a, hipStream_t s);
int vpt_attn_launch(const VptAttnArgs* a, hipStream_t s);
int vpt_kv_update_launch(const VptKvUpdateArgs* a, hipStream_t s);
int vpt_logsoftmax_launch(const VptLogSoftmaxArgs* a, hipStream_t s);
}
