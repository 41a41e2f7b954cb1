/* vpt_hip.h -- C ABI of libvpt_hip.so: the MI355X (gfx950) kernels behind the VPT policy hot path.
 *
 * The reference (openai/Video-Pre-Training) has no FFI of its own: its hot path is PyTorch ATen calls made
 * from lib/policy.py, lib/impala_cnn.py, lib/util.py, lib/xf.py, lib/masked_attention.py and
 * lib/action_head.py.  Each entry point below replaces one such group of calls (cited per function); the
 * Python host side (video-pre-training_amd/lib/policy.py) binds them with ctypes and keeps the reference's
 * MinecraftAgentPolicy API on top.  See INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (inputs, outputs, KV memory, workspaces);
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = default stream):
 *     no allocation, no synchronisation, no global state besides the last-error string;
 *   - return 0 on success, <0 on a rejected argument or a launch failure (vpt_last_error() explains);
 *   - "blocked" activations are bf16 [frames][C/32][H][W][32]; "stats" are double[frames][2] holding the
 *     running (sum, sum of squares) of a frame, accumulated with atomics -- the caller zeroes them;
 *   - packed weight formats are produced by the vpt_pack_* / vpt_chw_to_blocked entry points below from the
 *     reference's fp32 tensors: a host without torch can prepare a model (video-pre-training_amd/packing.py is the CPU restatement
 *     the tests compare them with, bit for bit).
 */
#ifndef VPT_HIP_H
#define VPT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library / build identification: returns "vpt_hip <version> gfx950". */
const char* vpt_version(void);
/* Binary-interface number of THIS header (VPT_HIP_ABI).  It is raised whenever an existing entry point changes its argument
 * list or a symbol is renamed or removed (adding entry points does not raise it); a caller compares it with the VPT_HIP_ABI it
 * was built against before the first call -- _native.py does -- instead of finding out through a mis-typed argument.
 * History: 3 = round 3 (vpt_adam_step_multi + skip_flag, vpt_heads_logprob_backward + grad_scale, vpt_gate_cast renamed);
 * 4 = round 4 (vpt_conv3x3_forward_tiled rejects tiling 0; vpt_action_head_forward takes a counter-based noise source). */
#define VPT_HIP_ABI 5
int vpt_abi_version(void);
/* "bf16" (libvpt_hip.so, the default) or "fp16" (libvpt_hip_f16.so: the same sources built with -DVPT_OPERAND_F16): the format
 * of every 16-bit buffer this library reads or writes -- activations, packed weights, MFMA operands.  Same ABI, same
 * MFMA rate; the fp16 build is the parity mode (8x finer operand rounding; lib/policy.py: precision="fp16"). */
const char* vpt_operand_format(void);
/* Human-readable reason for the most recent non-zero return on this thread. */
const char* vpt_last_error(void);

/* ---- weight re-packing (the `.weights` state_dict, fp32, reference shapes -> what the kernels stream) ------------------
 * A host needs nothing else to prepare a model: sizes from the queries, then one launch per layer.
 *   vpt_pack_conv3x3: Conv2d weight [Cout][Cin][3][3] of a FanInInitReLULayer with GroupNorm(1, Cin) (lib/util.py:58-82,
 *     lib/impala_cnn.py:30-52,86-97) -> wpk (16-bit, vpt_conv3x3_packed_elems elements: round(W * gain) in the swizzled LDS
 *     image of vpt_conv3x3_forward) and, when edge_sa / edge_sg are given (forward use; both or neither), the tables of the
 *     GroupNorm fold (vpt_conv3x3_table_floats floats each).  For vpt_conv3x3_dgrad pass the transposed, spatially flipped
 *     weight, gain = ones, no tables.
 *   vpt_pack_linear: nn.Linear weight [N][K] (row stride ldw) -> [ceil(N/128)][K/32][128][32] 16-bit, rows >= N zero,
 *     K % 64 == 0.  With transposed = 1 the packed matrix is the TRANSPOSE of a [src_rows][ldw] source (element (n, k) =
 *     source[k][n], zero for k >= src_rows): the operand of the input-gradient GEMM dx = dy W, with the reduction dimension
 *     (the layer's output width) padded to K.
 *   vpt_pack_conv_first: stack-0 firstconv weight [Cout][3][3][3] + bias [Cout] (lib/impala_cnn.py:86-97 with lib/util.py:64-65: no
 *     norm => bias) -> the MFMA A-operand fragments vpt_conv_first_forward / _backward read (16-bit, vpt_conv_first_packed_elems):
 *     W / 255 (the pixel operand is the raw byte; ImgPreprocessing's x / 255, lib/policy.py:39-45, is folded in here) and the bias
 *     as two 16-bit halves in the spare K slots.
 *   vpt_pack_conv3d_t5: IDM Conv3d weight [O][3][5][1][1] + bias [O] (lib/policy.py:364-372) -> fragments (16-bit,
 *     vpt_conv3d_t5_packed_elems) + the bias zero-padded to ceil(O/128)*128 floats.
 *   vpt_chw_to_blocked: fp32 [rows][C*H*W] in the reference's C,H,W flatten order (lib/impala_cnn.py:192-193) -> the blocked
 *     activation order [rows][C/32][H][W][32]: the dense layer's weight columns (rows = 256) and its LayerNorm gain / bias
 *     (rows = 1) before vpt_pack_linear / vpt_frame_affine_forward.  Out of place. */
long vpt_conv3x3_packed_elems(int Cout, int Cin);
long vpt_conv3x3_table_floats(int Cout);
long vpt_linear_packed_elems(int N, int K);
long vpt_conv_first_packed_elems(int Cout);
long vpt_conv3d_t5_packed_elems(int O);
int vpt_pack_conv_first(const float* weight, const float* bias, void* wfrag, int Cout, void* stream);
int vpt_pack_conv3d_t5(const float* weight, const float* bias, void* wfrag, float* bias_padded, int O, void* stream);
int vpt_chw_to_blocked(const float* src, float* dst, int64_t rows, int C, int H, int W, void* stream);

/* Workspace query: bytes of caller-owned scratch for the entry points that take a `scratch` / partial-sum buffer.
 *   VPT_WS_CONV3X3_WGRAD         (frames, -, -, Cin, Cout)   vpt_conv3x3_wgrad's scratch
 *   VPT_WS_CONV_BACKWARD_PREPARE (frames, -, -, -, Cout)     scratch of vpt_conv_backward_prepare / _prepare_pooled / _reduce
 *   VPT_WS_LINEAR_SPLITK         (splitk, M, N, -, -)        the [splitk][M][N] fp32 slices vpt_linear_forward writes when splitk > 1
 *   VPT_WS_LAYERNORM_BACKWARD    (M, D, -, -, -)             vpt_layernorm_backward's partials
 *   VPT_WS_COLUMN_SUM            (M, N, -, -, -)             vpt_column_sum's partials (0: none needed)
 *   VPT_WS_ATTENTION_BACKWARD_DKV  (B, t, hid, -, -)         vpt_masked_attention_backward's dkv_slab
 *   VPT_WS_ATTENTION_BACKWARD_DBND (B, t, heads, maxlen, -)  ... and its dbnd_slab
 *   VPT_WS_FRAME_AFFINE_BACKWARD (frames, HW, per_element, pass, C)   vpt_frame_affine_backward's partials for that pass
 *   VPT_WS_CONV_FIRST_BACKWARD   (frames, H, W, -, Cout)     vpt_conv_first_backward's partials (depends on the device's CU count)
 * Returns -1 for an unknown op. */
enum { VPT_WS_CONV3X3_WGRAD = 1, VPT_WS_CONV_BACKWARD_PREPARE = 2, VPT_WS_LINEAR_SPLITK = 3, VPT_WS_LAYERNORM_BACKWARD = 4, VPT_WS_COLUMN_SUM = 5,
       VPT_WS_ATTENTION_BACKWARD_DKV = 6, VPT_WS_ATTENTION_BACKWARD_DBND = 7, VPT_WS_FRAME_AFFINE_BACKWARD = 8, VPT_WS_CONV_FIRST_BACKWARD = 9 };
int64_t vpt_workspace_bytes(int op, int frames, int H, int W, int Cin, int Cout);
int vpt_pack_conv3x3(const float* weight, const float* gain, const float* bias, void* wpk, float* edge_sa, float* edge_sg,
                     int Cout, int Cin, void* stream);
int vpt_pack_linear(const float* weight, void* wpk, int N, int K, int transposed, int ldw, int src_rows, void* stream);

/* Stack-0 firstconv + ingest + ReLU + max-pool.
 * Replaces ImgPreprocessing.forward (lib/policy.py:39-45), the permute at lib/impala_cnn.py:190,
 * CnnDownStack.firstconv of stack 0 (lib/impala_cnn.py:86-97,115) and F.max_pool2d (lib/impala_cnn.py:117).
 * img: uint8 [frames][H][W][3]; wfrag: bf16 [NT][4][2][64][8]; y: blocked [frames][Cout/32][H/2][W/2][32].
 * out_gain (optional, [Cout], Cout <= 256): y is stored multiplied by it per channel -- the gain of the stack's GroupNorm `n` when that norm is
 * folded into the first block (vpt_nfold_coef); stats_out always holds the statistics of the UNscaled pooled tensor.
 * chs_out (optional, [frames][Cout][2] fp64, ACCUMULATED, Cout <= 128): per-channel (sum, sum of squares) of y as stored -- what
 * vpt_channel_stats would compute in a pass of its own; here the sums run in registers across the tiles of a frame. */
int vpt_conv_first_forward(const uint8_t* img, const void* wfrag, void* y, double* stats_out, const float* out_gain, double* chs_out,
                           int frames, int H, int W, int Cout, void* stream);

/* IDM temporal conv + ingest + bias + ReLU.
 * Replaces ImgPreprocessing.forward (lib/policy.py:39-45) and InverseActionNet._conv3d_forward
 * (lib/policy.py:394-403: Conv3d(3->Cout, kernel (5,1,1), padding (2,0,0)) over time, then ReLU).
 * img: uint8 [frames = B*T][H][W][3]; wfrag: bf16 [NT][4][64][8]; bias fp32 [NT*128];
 * y: blocked [frames][Cout/32][H][W][32]; stats_out (optional) receives the statistics of y. */
int vpt_conv3d_t5_forward(const uint8_t* img, const void* wfrag, const float* bias, void* y, double* stats_out,
                          int frames, int T, int H, int W, int Cout, void* stream);

/* GroupNorm(1,Cin) -> Conv2d(3x3, pad 1, no bias) -> ReLU [-> + residual].
 * Replaces FanInInitReLULayer.forward (lib/util.py:75-82) for conv layers and the residual add of
 * CnnBasicBlock.forward (lib/impala_cnn.py:50-52).  x, res, y blocked bf16; wpk bf16 [NT][Cin/32][9][128][32];
 * edge_sa / edge_sg fp32 [9][NT*128]; stats_in = statistics of x; stats_out (optional) receives those of y. */
int vpt_conv3x3_forward(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg,
                        const double* stats_in, const void* res, void* y, double* stats_out,
                        int frames, int H, int W, int Cin, int Cout, void* stream);
/* The same call with the workgroup tiling named by the caller: 1 = the throughput kernel (16x16 pixels x 128 output channels per
 * workgroup: batches of frames, what bench.py measures and what vpt_conv3x3_forward always runs), 2 = the latency kernel (x 32
 * output channels: 4x the workgroups, a quarter of the serial MFMA chain each -- the acting path of agent.py:190-206, where one
 * frame would otherwise occupy 2-32 of the 256 CUs), 3 = the throughput kernel on 32x16-pixel tiles with eight waves where H % 32 == 0
 * (one weight fetch per 512 pixels; bit-identical outputs, measured at parity with 1 -- kept for A/B measurements).  Any other value
 * is an error: the tiling is never derived from the grid size (a frame's result must not depend on how many frames share a launch).
 * Same layouts, same arithmetic, same K order. */
int vpt_conv3x3_forward_tiled(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg,
                              const double* stats_in, const void* res, void* y, double* stats_out,
                              int frames, int H, int W, int Cin, int Cout, int tiling, void* stream);

/* The same layer FUSED with the max-pool that follows it in CnnDownStack.forward (lib/impala_cnn.py:114-117: x = firstconv(x);
 * x = F.max_pool2d(x, 3, 2, 1)) -- stacks 1.., inference: pooled [frames][Cout/32][H/2][W/2][32] and its frame statistics come out, the
 * pre-pool tensor never reaches HBM.  Two launches inside: the convolution pools every 16 x 16 output tile through LDS and writes the
 * tile's last row / column to seam_scratch (vpt_conv3x3_pool_seam_elems(frames, H, W, Cout) 16-bit elements, caller-owned); a small
 * second kernel completes the pooled pixels whose 3 x 3 window crosses a tile border.  Bit-identical to vpt_conv3x3_forward followed by
 * vpt_maxpool_forward (the statistics to the order of their fp32 / fp64 additions).  No residual, throughput tiling only.
 * phases: 3 = both launches (the normal call); 1 = the convolution only, 2 = the seam kernel only (a caller that times them apart).
 * out_gain, chs_out: as for vpt_conv_first_forward (the pooled tensor stored times GroupNorm `n`'s gain, statistics of the unscaled
 * values; per-channel sums of the stored tensor, accumulated by both launches). */
int64_t vpt_conv3x3_pool_seam_elems(int frames, int H, int W, int Cout);
int vpt_conv3x3_pool_forward(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg, const double* stats_in,
                             void* pooled, void* seam_scratch, double* stats_out, const float* out_gain, double* chs_out, int frames, int H,
                             int W, int Cin, int Cout, int phases, void* stream);

/* Round 5, the TRAINING forward of a stack's firstconv -> max_pool2d (lib/impala_cnn.py:114-117 under behavioural_cloning.py:101): the same pass
 * that ALSO records, per pooled value, which positions of its 3 x 3 window hold the maximum -- pool_mask [F][Cout/32][H/2][W/2][32] uint16, bit
 * 8 - k set = scan position k = 3 (dy + 1) + (dx + 1) differs from the maximum or lies outside the image.  The backward
 * (vpt_conv_backward_prepare_pooled) routes the pooled gradient to the first zero bit -- torch's first-maximum rule -- so neither the
 * pre-pool tensor (2 MB per frame in stack 1) nor vpt_maxpool_forward's arg-max bytes exist in the BC step.  Pooled values and statistics
 * are those of vpt_conv3x3_pool_forward, bit for bit.  phases as there. */
int vpt_conv3x3_pool_argmax_forward(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg, const double* stats_in,
                                    void* pooled, void* pool_mask, void* seam_scratch, double* stats_out, int frames, int H, int W, int Cin, int Cout, int phases, void* stream);

/* ---- GroupNorm `n` of a stack folded into its first residual block (CnnDownStack.forward, lib/impala_cnn.py:118-121: x = self.n(x);
 * for block in self.blocks: x = block(x)) -- inference.  The producer of the pooled tensor P stores Q = n.weight[c] * P (out_gain above);
 * x = n(P) = r_P Q + b[c] is never written:
 *   vpt_channel_stats: chs[f][c] = (sum_p Q, sum_p Q^2) over the HW pixels of a frame, fp64, ACCUMULATED (caller zeroes);
 *   vpt_nfold_coef:    from tot[f] = the frame statistics of P (the producer's stats_out), chs, n.weight / n.bias and four edge tables of
 *                      block 0's conv0 -- its own edge_sa / edge_sg plus tb / tg = the edge_sg sums with every input channel weighted by
 *                      n.bias / n.weight -- the per-frame epilogue table kk_frame [frames][9][ceil(Cout/128)*128], the accumulator scale
 *                      rs_frame [frames] (conv0) and res_scale [frames], res_bias [frames][C] (conv1's residual x = res_scale * Q + res_bias);
 *   vpt_conv3x3_forward_folded: vpt_conv3x3_forward with either of the two substitutions: (kk_frame, rs_frame) replace edge_sa and the
 *                      statistics of x (conv0 on Q; edge_sg must still point at a table of the right size, stats_in may be null);
 *                      (res_scale, res_bias) turn the residual into res_scale[f] * res + res_bias[f][c] (conv1, res = Q).
 * Saves the read and the write of every pooled tensor that vpt_frame_affine_forward costs. */
int vpt_channel_stats(const void* x, double* chs, int frames, int C, int HW, void* stream);
int vpt_nfold_coef(const double* tot, const double* chs, const float* gain, const float* bias, const float* sa, const float* sg,
                   const float* tb, const float* tg, float* kk_frame, float* rs_frame, float* res_scale, float* res_bias,
                   int frames, int C, int HW, int Cout, void* stream);
int vpt_conv3x3_forward_folded(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg, const double* stats_in,
                               const float* kk_frame, const float* rs_frame, const void* res, const float* res_scale, const float* res_bias,
                               void* y, double* stats_out, int frames, int H, int W, int Cin, int Cout, void* stream);

/* F.max_pool2d(x, 3, 2, 1) on a post-ReLU blocked tensor (lib/impala_cnn.py:117, stacks 1..2).  argmax (optional, for
 * training): uint8, shaped like y, the window position kh*3+kw of the first maximum (15 when the window is all zero). */
int vpt_maxpool_forward(const void* x, void* y, double* stats_out, uint8_t* argmax, int frames, int C, int H, int W, void* stream);

/* y = (x - mean_f) * rstd_f * gain + bias with whole-frame statistics:
 * CnnDownStack.n (GroupNorm(1,C), lib/impala_cnn.py:99-100,118-119; per_element = 0, gain[C]) and the
 * LayerNorm of ImpalaCNN.dense over the flattened frame (lib/impala_cnn.py:177-184; per_element = 1,
 * gain[C*H*W] already permuted to the blocked order). */
int vpt_frame_affine_forward(const void* x, void* y, const float* gain, const float* bias,
                             const double* stats_in, double* stats_out,
                             int frames, int C, int HW, int per_element, void* stream);

/* C[M,N] = A[M,K] W[N,K]^T (+bias) (ReLU) (+res): every nn.Linear on the path (lib/util.py:58-82,
 * lib/xf.py:251-254, lib/action_head.py:164, lib/scaled_mse_head.py:35).  A bf16 [M][lda]; wpk bf16
 * [ceil(N/128)][K/32][128][32]; bias fp32[N] or NULL; res fp32 [M][ldr] or NULL; out_f32 [M][ldc] and/or
 * out_bf16 [M][ldcb].  splitk > 1 (no ReLU/res): out_f32 is a caller-zeroed [splitk][M][ldc] buffer, split s writes its
 * partial product to slice s and the caller sums the slices -- no atomics, so the result is bit-reproducible.
 * mask (optional, bf16 [M][ldm]) zeroes outputs where mask <= 0 before the residual add: the ReLU backward of
 * the BC step's dgrad GEMMs.  The same entry point serves forward, dgrad (W^T packed) and wgrad (A = dY^T). */
int vpt_linear_forward(const void* A, const void* wpk, const float* bias, const float* res,
                       float* out_f32, void* out_bf16, int M, int N, int K,
                       int lda, int ldr, int ldc, int ldcb, int relu, int splitk, const void* mask, int ldm,
                       void* stream);
/* The same with the kernel named by the caller: 1 = the MFMA GEMM whatever M, 2 = the weight-streaming kernel of the acting step (M <= 8
 * rows), 0 = by M (what vpt_linear_forward does: <= 8 rows stream the weights).  The two kernels sum in different orders; a caller
 * whose rows must not depend on how many rows share the call (batches of sequences: chunking, sharding) passes 1.  4 = the MFMA GEMM on
 * 256 x 256 tiles with eight waves and LDS-DMA operands where that grid fills the chip (bit-identical to 1; measured neutral inside the
 * engine, kept for A/B measurements). */
int vpt_linear_forward_tiled(const void* A, const void* wpk, const float* bias, const float* res,
                             float* out_f32, void* out_bf16, int M, int N, int K,
                             int lda, int ldr, int ldc, int ldcb, int relu, int splitk, const void* mask, int ldm,
                             int tiling, void* stream);

/* Weight gradient of a linear layer without transposed copies: dw[n][k] (+)= sum_m dy[m][n] * x[m][k]; dy bf16 [M][ldy],
 * x bf16 [M][ldx] (both row-major over the M frames / tokens), dw fp32 [N][ldw]; N, K, ldy, ldx multiples of 8.
 * Replaces autograd's dW = dY^T X for every nn.Linear of the BC step (behavioural_cloning.py:117-119). */
int vpt_linear_wgrad(const void* dy, const void* x, float* dw, int M, int N, int K, int ldy, int ldx, int ldw, int accumulate, void* stream);

/* ImpalaCNN.dense with its LayerNorm folded into the GEMM (lib/impala_cnn.py:177-194; inference): run vpt_linear_forward with splitk > 1 on the RAW
 * block output x [M = frames][K = C*16*16] against weights packed from W * gain (per element), then this: out[f][n] = rstd_f * sum_s part[s][f][n]
 * - rstd_f * mean_f * sg[n] + sb[n], (mean_f, rstd_f) from stats [M][2] of x (count = K), sg[n] = sum_k op16(W gain)[n][k], sb[n] = sum_k W[n][k]
 * bias[k].  Replaces vpt_frame_affine_forward(per_element = 1) + the split-K sum. */
int vpt_dense_fold_epilogue(const float* part, int splitk, const double* stats, int count, const float* sg, const float* sb, float* out,
                            int M, int N, void* stream);

/* Second stage of a split-K linear: out = epilogue(sum_s part[s][M][N]) with the same epilogue options as
 * vpt_linear_forward (part = the [splitk][M][N] buffer a vpt_linear_forward call with splitk > 1 filled). */
int vpt_linear_splitk_epilogue(const float* part, int splitk, const float* bias, const float* res, float* out_f32, void* out_bf16,
                               int M, int N, int ldr, int ldc, int ldcb, int relu, const void* mask, int ldm, void* stream);

/* Acting path (M = B*T <= 8 rows, K <= 3072; agent.py:190-206): nn.LayerNorm (optional ReLU on its input) FUSED into the linear
 * layer it feeds -- every LayerNorm of the policy does feed one (lib/xf.py:334-356 pre_r_ln -> q/k/v/r, ln -> mlp0; lib/util.py:58-82
 * FanInInitReLULayer(norm, linear); lib/policy.py:188,211-214 lastlayer, final_ln -> heads).  Same results, bit for bit, as
 * vpt_layernorm_forward followed by vpt_linear_forward; ln_out_f32 (optional, [M][K]) receives the normalised rows in fp32
 * (the residual branch / the latent).  Larger M or K: returns -1, call the two functions. */
int vpt_layernorm_linear_forward(const float* x, const float* ln_gain, const float* ln_bias, int relu_in, float* ln_out_f32,
                                 const void* wpk, const float* bias, const float* res, float* out_f32, void* out_bf16,
                                 int M, int N, int K, int ldr, int ldc, int ldcb, int relu, void* stream);

/* nn.LayerNorm over the last dim with optional ReLU on the input (lib/util.py:61-62,169; lib/policy.py:188,211-214). */
int vpt_layernorm_forward(const float* x, const float* gain, const float* bias, float* out_f32, void* out_bf16,
                          int M, int D, int relu_in, void* stream);

/* Attention of MaskedAttention.forward (lib/masked_attention.py:161-178) over attention() (lib/xf.py:18-71).
 * causal = 1 ("clipped_causal"): banded-causal with KV memory and relative-position bias; qkvr fp32 [B*t][ld] =
 * Q | K | V | R columns; kmem/vmem fp32 [B][maxlen][hid]; memvalid uint8 [B][maxlen] (= state_mask &
 * !first[:,0]); b_nd fp32 [10][maxlen].  causal = 0 (mask "none", the IDM): maxlen must be 0 (no memory; the
 * rel-pos bias is identically zero, lib/util.py:256-260), every query attends to all t <= 160 rows of its
 * chunk; kmem/vmem/memvalid/b_nd are ignored.  out bf16 [B*t][hid] (heads merged, lib/xf.py:125-131). */
int vpt_masked_attention_forward(const float* qkvr, const float* kmem, const float* vmem, const uint8_t* memvalid,
                                 const float* b_nd, void* out, int B, int t, int heads, int hid, int ld,
                                 int maxlen, int causal, void* stream);

/* SelfAttentionLayer.update_state (lib/xf.py:366-391): kout/vout = last maxlen rows of [memory ; new]. */
int vpt_kv_memory_update(const float* qkvr, const float* kmem, const float* vmem, float* kout, float* vout,
                         int B, int t, int hid, int ld, int maxlen, void* stream);

/* Acting step (t = 1; agent.py:190-206): the whole recurrent-state step of one transformer block in ONE launch --
 * vpt_masked_attention_forward(causal = 1, t = 1), vpt_kv_memory_update and the mask bookkeeping of lib/xf.py:366-391
 * (memory visible where state_mask & ~first; next mask = cat(state_mask[1:] & ~first, [True])).  One query per (sequence, head)
 * against the newest maxlen - 1 memory rows and the token itself; the rows it reads are written one row up into kout / vout.
 * state_mask / mask_out: bool bytes [B][maxlen]; first: bool bytes [B].  kout / vout MAY alias kmem / vmem (in-place state, as
 * the captured acting graph uses it); mask_out must be a different buffer than state_mask.  maxlen <= 128.  Same formulas as the general kernel
 * (lib/xf.py:18-71, lib/masked_attention.py:161-178); sums in a different order: equal to fp32 rounding. */
int vpt_masked_attention_step(const float* qkvr, const float* kmem, const float* vmem, const uint8_t* state_mask, const uint8_t* first,
                              const float* b_nd, void* out, float* kout, float* vout, uint8_t* mask_out,
                              int B, int heads, int hid, int ld, int maxlen, void* stream);

/* The same step with EVERYTHING updated in place (what the captured acting graph uses: no copies of the recurrent state at all):
 * kmem / vmem / state_mask are read and overwritten.  The mask is safe to overwrite because the workgroup that arrives last at
 * done_counter ([B] ints, zero before the first launch; the kernel leaves them zero) writes it, after every other one has read it. */
int vpt_masked_attention_step_inplace(const float* qkvr, float* kmem, float* vmem, uint8_t* state_mask, const uint8_t* first,
                                      const float* b_nd, void* out, int* done_counter, int B, int heads, int hid, int ld, int maxlen, void* stream);

/* Tail of MinecraftAgentPolicy.act for the acting step (lib/policy.py:307-327) in one launch: log_prob = logp_buttons + logp_camera
 * (lib/action_head.py:227-237 sums the heads), value de-normalisation v * scale + shift (lib/normalize_ewma.py:27-31 with
 * scale = sqrt(var + eps), shift = mean), the NaN assertion of lib/policy.py:320-321 as a byte flag, and one packed record per
 * environment:  keep[b][4] (int64) = { buttons action, camera action, float bits of log_prob, float bits of the de-normalised value
 * (low half) | float bits of the raw value-head output (high half) }.  value_col: column of the value head in logits[B][ld].
 * rng_state (optional, device uint64 {seed, step}): step += 1 -- the stochastic heads of this acting step have drawn (vpt_action_head_forward). */
int vpt_act_epilogue(const int64_t* action_buttons, const int64_t* action_camera, const float* logp_buttons, const float* logp_camera,
                     const float* logits, int ld, int value_col, float scale, float shift, int64_t* keep, uint8_t* nan_flag, uint64_t* rng_state,
                     int B, void* stream);

/* CategoricalActionHead.forward tail (lib/action_head.py:170-174): out[M][n] = log_softmax(logits[:, col0:col0+n] / T). */
int vpt_log_softmax_forward(const float* logits, float* out, int M, int ld, int col0, int n, float temperature,
                            void* stream);

/* The whole CategoricalActionHead on one head's columns (lib/action_head.py:163-207): log-probs as above, with
 *   mask   [M][n] uint8, optional: 0 = unavailable action, its scaled logit becomes LOG0 = -100 before the softmax
 *          (shaped_out[~mask] = LOG0, lib/action_head.py:170-171; obs["mask"] of lib/policy.py:257-266);
 *   action [M] int64, optional: CategoricalActionHead.sample -- arg-max of the log-probs (deterministic), or of
 *          log-probs - log(-log u) with u = noise[M][n] uniform in [0,1] (Gumbel-max, u == 1 -> 0.999 as the reference);
 *          the FIRST maximum, as torch.argmax.  The random numbers are the caller's (noise; the reference draws them from torch's
 *          generator, lib/action_head.py:200) OR generated inside the kernel from rng_state = device uint64 {seed, step} and
 *          rng_stream (Philox4x32-10: key = seed, counter = (element / 4, row, step, stream << 24), u = (word >> 8) * 2^-24 --
 *          what a captured acting step needs: a fresh draw per replay with no host generator in the loop).  The kernel does not
 *          advance `step`: vpt_act_epilogue does, once per acting step (or the host between calls).  noise and rng_state are
 *          mutually exclusive; both null = deterministic;
 *   action_logp [M] fp32, optional: log-prob of that action (CategoricalActionHead.logprob, lib/action_head.py:176-184). */
int vpt_action_head_forward(const float* logits, const uint8_t* mask, const float* noise, const uint64_t* rng_state, uint32_t rng_stream,
                            float* out, int64_t* action, float* action_logp, int M, int ld, int col0, int n, float temperature, void* stream);
/* out[M][n] = exactly the uniforms vpt_action_head_forward generates for (rng_state, rng_stream) -- to inspect or replay a draw through
 * the `noise` argument (th.rand_like(logits), lib/action_head.py:200).  rng_state is not advanced. */
int vpt_uniform_noise(const uint64_t* rng_state, uint32_t rng_stream, float* out, int M, int n, void* stream);

/* Fused Adam update of one flat fp32 bucket: th.optim.Adam(lr, weight_decay).step() as configured by
 * behavioural_cloning.py:63-67,122 (L2 weight decay folded into the gradient, bias-corrected moments).
 * `step` counts from 1; grad_scale multiplies the gradient first (1/world_size of the data-parallel mean). */
int vpt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);

/* The same update for EVERY parameter tensor in one launch (th.optim.Adam(policy.parameters()).step(), behavioural_cloning.py:63-67,
 * 122).  `table` is a DEVICE array of `ntensors` descriptors { float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
 * uint64_t n; int64_t first_block; } (48 bytes each), sorted by first_block, where first_block counts 1024-element blocks:
 * first_block[0] = 0, first_block[i+1] = first_block[i] + ceil(n[i] / 1024); total_blocks = the sum.
 * skip_flag (device int32, optional): when non-zero the launch leaves every tensor untouched -- the found-inf result of
 * vpt_grads_nonfinite_multi for a loss-scaled step (fp16 operand format), consumed without a host round trip. */
int vpt_adam_step_multi(const void* table, int ntensors, int64_t total_blocks, int step, float lr, float beta1, float beta2,
                        float eps, float weight_decay, float grad_scale, const int32_t* skip_flag, void* stream);

/* Sets *flag = 1 (device int32, zeroed by the caller) if any gradient of the table (same layout as above; only `grad`, `n`,
 * `first_block` are read) is inf or nan.  The reference trains in fp32 and has no counterpart; this is what
 * torch.cuda.amp.GradScaler.unscale_ computes for an fp16 run of behavioural_cloning.py:117-122. */
int vpt_grads_nonfinite_multi(const void* table, int ntensors, int64_t total_blocks, int32_t* flag, void* stream);

/* ---- behavioural-cloning step: backward of the heads / trunk / transformer (the reference uses torch autograd,
 * behavioural_cloning.py:117-119).  Linear layers reuse vpt_linear_forward (dgrad: W^T packed; wgrad: A = dY^T). */

/* d loss / d logits for loss = -sum_rows[ log_softmax(z_b/T)[a_b] + log_softmax(z_c/T)[a_c] ] * (scale*T), written as
 * bf16 [M][ldz] (columns >= nb+nc zero): scale = 1 / (global frames * temperature).  lib/action_head.py:170-184. */
int vpt_bc_nll_backward(const float* lp_buttons, const float* lp_camera, const int64_t* act_buttons,
                        const int64_t* act_camera, void* dz, int M, int nb, int nc, int ldz, float scale, void* stream);

/* The same boundary for an ARBITRARY incoming gradient -- what torch autograd hands to the outputs of
 * MinecraftAgentPolicy.forward / get_output_for_observation (lib/policy.py:252-305) when the caller writes its own loss, as
 * behavioural_cloning.py:101-119 does: g_buttons / g_camera = d loss / d log-prob ([M][nb] / [M][nc], either may be NULL),
 * g_value = d loss / d (raw value-head output) ([M], may be NULL).  dz = (g - exp(lp) * rowsum(g)) / temperature per head,
 * the value column passes through; 16-bit [M][ldz], ldz >= nb + nc + 1, padding zero.  grad_scale multiplies every written value
 * (the loss scale of the fp16 operand format, whose 16-bit gradient buffers would otherwise underflow; 1 for bf16).  mask_* (uint8 [M][n], optional): the
 * availability masks of the forward -- a masked logit was overwritten with the constant LOG0, so its dz is 0. */
int vpt_heads_logprob_backward(const float* lp_buttons, const float* lp_camera, const float* g_buttons, const float* g_camera,
                               const float* g_value, const uint8_t* mask_buttons, const uint8_t* mask_camera, void* dz,
                               int M, int nb, int nc, int ldz, float temperature, float grad_scale, void* stream);

/* nn.LayerNorm backward (optionally through a ReLU on the LayerNorm's input): dx = dx_add + dLN(x, dy);
 * dgain / dbias += the column sums (caller zeroes), reduced in a FIXED order through `partials` (fp32 workspace of
 * vpt_workspace_bytes(VPT_WS_LAYERNORM_BACKWARD, M, D, ...) bytes): the same inputs give the same bits, as torch autograd on one device does for
 * behavioural_cloning.py:117-119.  (ABI 5; ABI 4 accumulated with fp32 atomics in arrival order.) */
int vpt_layernorm_backward(const float* x, const float* gain, const float* dy, const float* dx_add, float* dx,
                           float* dgain, float* dbias, float* partials, int M, int D, int relu_in, void* stream);

/* out16[M][ldo] = (mask > 0 ? x : 0) with columns >= N zeroed: the ReLU backward gate (F.relu at lib/util.py:81,
 * lib/policy.py:211) fused with the cast / K-padding that turns an fp32 gradient into a GEMM A operand. */
int vpt_gate_cast(const float* x, const void* mask, void* out, int M, int N, int ldx, int ldm, int ldo, void* stream);

/* out[N] += column sums of a 16-bit [M][ld] matrix (bias gradients), in a fixed order through `partials` (VPT_WS_COLUMN_SUM (M, N) bytes; may be
 * NULL when that is 0). */
int vpt_column_sum(const void* x_bf16, float* out, float* partials, int M, int N, int ld, void* stream);

/* Backward of vpt_masked_attention_forward (causal = 1): dqkvr [B*t][ld] receives dQ, dK, dV and dR (every column written, nothing to zero);
 * db_nd [10][maxlen] accumulated (caller zeroes).  The KV memory is detached state (behavioural_cloning.py:111) and gets no gradient.
 * A key is reached by up to five 32-query tiles and b_nd by every workgroup: each contribution is written to its own slot of a workspace and the
 * slots are added in a fixed order (bit-reproducible).  dkv_slab: VPT_WS_ATTENTION_BACKWARD_DKV (B, t, hid) bytes, dbnd_slab:
 * VPT_WS_ATTENTION_BACKWARD_DBND (B, t, heads, maxlen) bytes.  ld must be a multiple of 4. */
int vpt_masked_attention_backward(const float* qkvr, const float* kmem, const float* vmem, const uint8_t* memvalid,
                                  const float* b_nd, const float* dout, float* dqkvr, float* db_nd, float* dkv_slab, float* dbnd_slab,
                                  int B, int t, int heads, int hid, int ld, int maxlen, void* stream);

/* ---- backward of the IMPALA CNN (behavioural_cloning.py:117-119 obtains these from torch autograd) ---- */

/* Per-element preparation of GN -> conv3x3 -> ReLU (+res) backward: dacc = rstd * dY * [v > 0] (blocked, Cout
 * channels); t12[f] = (T1, T2) = (sum dz (v - SA), sum dz SG) (optional output); coef[f] = (c0, c1) with
 * c1 = -rstd^2 T1 / n, c0 = -rstd T2 / n - c1 mu, the statistics terms vpt_conv3x3_dgrad adds as c0 + c1 x;
 * d_sa / d_sg [9][CoutPad] += sum dz / sum dz (-rstd mu) (accumulated: caller zeroes once per step).
 * stats_in are the statistics of the layer's INPUT (Cin*H*W elements).  With dy = NULL the layer is followed by the
 * max-pool and (dpooled, argmax) are given instead: the pool's backward is applied on the fly.  scratch: fp32 work buffer of
 * vpt_workspace_bytes(VPT_WS_CONV_BACKWARD_PREPARE, frames, 0, 0, 0, Cout) bytes (per-frame sums, then per-32-frame partial sums of d_sa / d_sg that
 * are added in a fixed order).  W must be 8, 16, 32 or 64. */
int vpt_conv_backward_prepare(const void* dy, const void* dpooled, const uint8_t* argmax, const void* y, const void* res,
                              const double* stats_in, const float* edge_sa, const float* edge_sg, void* dacc, double* t12,
                              float* coef, float* d_sa, float* d_sg, float* scratch, int frames, int H, int W, int Cin, int Cout, void* stream);

/* Input gradient of the layer: dx = conv^T(W', dacc) + skip + coef[f][0] + coef[f][1] * xin, i.e. the implicit-GEMM
 * kernel of vpt_conv3x3_forward on the transposed, spatially flipped weights (wpk_t: [ceil(Cin/128)][Cout/32][9][128][32])
 * with the GroupNorm-statistics terms c0_f + c1_f x added in its epilogue.  dacc has Cout channels, dx / xin / skip Cin.
 * xin and coef are required (rejected with -1 otherwise); skip may be NULL. */
int vpt_conv3x3_dgrad(const void* dacc, const void* wpk_t, const void* skip, const void* xin, const float* coef, void* dx,
                      int frames, int H, int W, int Cout, int Cin, void* stream);

/* Round 5: a block's conv1 -> conv0 backward without the per-element pass in between (lib/impala_cnn.py:50-52 under
 * behavioural_cloning.py:117-119).  vpt_conv3x3_dgrad_gated is vpt_conv3x3_dgrad (no skip) whose epilogue applies the NEXT layer-to-
 * differentiate's ReLU gate and statistic scale: xin is conv0's output y (conv1's input), conv0 has no residual, so
 *     dacc_out = rstd_g[f] * (conv^T(W', dacc) + c0 + c1 xin) * [xin > 0]
 * IS conv0's backward operand (what vpt_conv_backward_prepare(dy, y, res = NULL) would have written), rstd_g from gate_stats = the frame
 * statistics of conv0's INPUT over gate_cin * H * W elements; gate_u[f] += rstd_g * sum dy * xin (accumulated: caller zeroes).
 * vpt_conv_backward_reduce then produces what prepare produces besides dacc -- t12, coef, d_sa / d_sg -- from dacc_out and gate_u alone
 * (one read of one tensor instead of two reads and a write).  scratch as for vpt_conv_backward_prepare. */
int vpt_conv3x3_dgrad_gated(const void* dacc, const void* wpk_t, const void* xin, const float* coef, const double* gate_stats, int gate_cin,
                            void* dacc_out, double* gate_u, int frames, int H, int W, int Cout, int Cin, void* stream);
int vpt_conv_backward_reduce(const void* dacc, const double* gate_u, const double* stats_in, const float* edge_sa, const float* edge_sg,
                             double* t12, float* coef, float* d_sa, float* d_sg, float* scratch, int frames, int H, int W, int Cin, int Cout, void* stream);

/* vpt_conv_backward_prepare for the layer in front of the max-pool when its forward was vpt_conv3x3_pool_argmax_forward: (dpooled, pooled,
 * pool_mask) [F][Cout/32][H/2][W/2][32] in, dacc [F][Cout/32][H][W][32] and the same sums out.  The ReLU gate is [pooled > 0], the value at
 * the arg-max is the pooled value itself.  H, W: the PRE-pool size (W in {16, 32, 64}).
 * n_gain != NULL: the stack's GroupNorm `n` (lib/impala_cnn.py:118-119) sits between the pool and the incoming gradient and its backward is applied
 * on the fly -- `dpooled` is then G = d loss / d n(pooled), pool_stats [F][2] the frame statistics of pooled and pool_ab [F][2] the sums of
 * vpt_frame_affine_backward's pass 1 (sum G gain, sum G gain xhat): d(pooled) = r (G gain - ab0 / n - xhat ab1 / n) is formed per element with the
 * separate pass's arithmetic and 16-bit rounding point, and vpt_frame_affine_backward's pass 2 is not run for this stack. */
int vpt_conv_backward_prepare_pooled(const void* dpooled, const void* pooled, const void* pool_mask, const double* stats_in, const float* edge_sa, const float* edge_sg,
                                     void* dacc, double* t12, float* coef, float* d_sa, float* d_sg, float* scratch,
                                     const float* n_gain, const double* pool_stats, const double* pool_ab, int frames, int H, int W, int Cin, int Cout, void* stream);

/* Backward of vpt_conv_first_forward w.r.t. its weight and bias (the input is the uint8 image): recomputes the pre-pool
 * tile, routes dpooled to the arg-max conv pixel of every pooling window and accumulates dw[Cout][27] (kh, kw, ch order)
 * and db[Cout] (caller zeroes) -- per-workgroup partial sums in `partials` (VPT_WS_CONV_FIRST_BACKWARD (frames, H, W, -, Cout) bytes), added in
 * a fixed order. */
int vpt_conv_first_backward(const uint8_t* img, const void* wfrag, const void* dpooled, float* dw, float* db, float* partials,
                            int frames, int H, int W, int Cout, void* stream);
/* ... with the stack's GroupNorm `n` backward (lib/impala_cnn.py:118-119) applied on the fly, as vpt_conv_backward_prepare_pooled does for stacks 1..:
 * `g` is d loss / d n(pooled), n_gain [Cout] the norm's gain, pool_stats [F][2] the frame statistics of the pooled tensor and pool_ab [F][2] the sums of
 * vpt_frame_affine_backward's pass 1.  d(pooled) = r (g gain - ab0 / n - xhat ab1 / n) is formed per element with the separate pass's arithmetic and 16-bit
 * rounding point; the pooled VALUE it needs is the window maximum the kernel's arg-max search finds anyway.  vpt_frame_affine_backward's pass 2 is not run
 * for stack 0. */
int vpt_conv_first_backward_nfold(const uint8_t* img, const void* wfrag, const void* g, const float* n_gain, const double* pool_stats, const double* pool_ab,
                                  float* dw, float* db, float* partials, int frames, int H, int W, int Cout, void* stream);

/* Weight gradient of the folded convolution: dw[o][tap][c] += sum_{f,p} dacc[f][o][p] * x[f][c][p + tap] (fp32; caller
 * zeroes or accumulates).  W in {16, 32, 64}.  scratch: fp32 work buffer of vpt_conv3x3_wgrad_scratch_floats() elements
 * (per-frame-group partial sums).  The host maps dw to dW, dgain, dbias (training.py). */
long vpt_conv3x3_wgrad_scratch_floats(int frames, int Cin, int Cout);
int vpt_conv3x3_wgrad(const void* dacc, const void* x, float* dw, float* scratch, int frames, int H, int W, int Cin, int Cout, void* stream);

/* F.max_pool2d(3, 2, 1) backward with torch's first-maximum tie rule. */
int vpt_maxpool_backward(const void* pre, const void* pooled, const void* dpooled, void* dpre,
                         int frames, int C, int H, int W, void* stream);

/* Backward of vpt_frame_affine_forward.  pass 1: ab[f] += (sum dy g, sum dy g xhat) and (per_element = 0) dgain/dbias;
 * pass 2: dx = rstd (dy g - ab0/n - xhat ab1/n) + dx_add; pass 3 (per_element = 1): dgain/dbias reduced over frames.
 * partials: fp32 workspace of VPT_WS_FRAME_AFFINE_BACKWARD (frames, HW, per_element, pass, C) bytes (0 for pass 2 and for pass 1 with
 * per_element = 1: may be NULL there): dgain / dbias are summed through it in a fixed order. */
int vpt_frame_affine_backward(const void* x, const void* dy, const void* dx_add, void* dx, const float* gain,
                              const double* stats_in, double* ab, float* dgain, float* dbias, float* partials,
                              int frames, int C, int HW, int per_element, int pass, void* stream);

/* ---- action codec on the device: int64 / fp64 arrays, one row per action ---- */

/* CameraQuantizer.discretize / undiscretize (lib/actions.py:88-108): n scalars (both camera axes flattened); mu_law = 1
 * selects the mu-law companding of agent.py:40-45 (maxval 10, binsize 2, mu 10), 0 the linear scheme.  fp64 arithmetic,
 * round-half-to-even like np.round. */
int vpt_camera_discretize(const double* xy, long* bins, long n, double maxval, double binsize, double mu, int mu_law, void* stream);
int vpt_camera_undiscretize(const long* bins, double* xy, long n, double maxval, double binsize, double mu, int mu_law, void* stream);

/* CameraHierarchicalMapping.from_factored / to_factored (lib/action_mapping.py:179-219): buttons int64 [n][20] in
 * Buttons.ALL order (lib/actions.py:21-33), camera int64 [n][2] bins <-> joint_buttons int64 [n] in 0..8640 and
 * joint_camera int64 [n] in 0..n_camera_bins^2-1.  Inputs are not validated (the reference raises KeyError on bins
 * outside the grid). */
int vpt_action_from_factored(const long* buttons, const long* camera, long* joint_buttons, long* joint_camera, long n, int n_camera_bins, void* stream);
int vpt_action_to_factored(const long* joint_buttons, const long* joint_camera, long* buttons, long* camera, long n, int n_camera_bins, void* stream);

/* ---- clip data path on the device ---- */

/* One launch turns `frames` decoded video frames (uint8 BGR [frames][height][width][3], as cv2.VideoCapture.read() returns
 * them) into the policy's input frames (uint8 RGB [frames][out_height][out_width][3]): the per-frame body of
 * data_loader.py:113-122 = composite_images_with_alpha (data_loader.py:34-46) where cursor_state[f] = (gui open, x, y) says a
 * GUI is open (x, y >= 0: the cursor's top-left corner, already scaled by height / 720), cv2.cvtColor(BGR2RGB) and
 * resize_image = cv2.resize(.., (out_width, out_height), INTER_LINEAR) (agent.py:100-103).  Bit-identical to that CPU path:
 * fp64 blend truncated like astype(uint8); OpenCV's 11-bit fixed-point bilinear (exact 2 x 2 decimation -> INTER_AREA).
 * cursor_state may be null (no compositing); cursor_bgr [cursor_h][cursor_w][3], cursor_alpha fp64 [cursor_h][cursor_w] in 0..1. */
int vpt_clip_frames(const uint8_t* src_bgr, int frames, int height, int width, const int32_t* cursor_state, const uint8_t* cursor_bgr,
                    const double* cursor_alpha, int cursor_h, int cursor_w, uint8_t* dst_rgb, int out_height, int out_width, void* stream);

/* ---- diagnostics ---- */

/* Fill the LDS of every compute unit with NaN bit patterns (0x7fc07fc0: a NaN as fp32 and as either 16-bit operand format), enqueued on
 * `stream`.  A kernel that reads LDS before writing it then produces NaN instead of whatever the previous workgroup on that CU left behind
 * (tools/diag_r06.py, VPT_POISON_LDS=1 in ops.py).  No reference counterpart; never on a product path. */
int vpt_debug_poison_lds(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VPT_HIP_H */
