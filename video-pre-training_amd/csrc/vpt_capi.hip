// C ABI (include/vpt_hip.h) on top of the per-kernel launchers.  Plain pointers and sizes only.
#include "../../include/vpt_hip.h"
#include "vpt_kernels.h"
#include "vpt_common.h"
#include <stdio.h>
#include <math.h>

static thread_local char g_err[256] = "ok";

static int fail(int code, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s (code %d, hip: %s)", what, code, hipGetErrorString(hipPeekAtLastError()));
  return code;
}
#define CHECK_LAUNCH(expr, name) do { int rc_ = (expr); if (rc_ != 0) return fail(rc_, name); return 0; } while (0)

// ---- diagnostics: fill the LDS of every CU with NaN patterns (vpt_debug_poison_lds) ---------------------------------------------
// A kernel that reads LDS it has not written sees whatever the previous workgroup on that CU left there: benign while one process
// repeats one kernel sequence, different when another process's workgroups interleave.  Launched before a kernel under test, this
// makes such a read show as NaN at once.  One workgroup takes a CU's whole 160 KB and lingers, so that a launch of many covers all CUs.
__global__ __launch_bounds__(256) void vpt_poison_lds_kernel(int words, unsigned pattern) {
  extern __shared__ unsigned poison_lds_[];
  volatile unsigned* l = poison_lds_;
  for (int i = threadIdx.x; i < words; i += 256) l[i] = pattern;
  __syncthreads();
  for (int k = 0; k < 16; ++k) __builtin_amdgcn_s_sleep(127);
}

extern "C" {

const char* vpt_version(void) { return "vpt_hip 0.5 gfx950"; }
int vpt_abi_version(void) { return VPT_HIP_ABI; }
const char* vpt_operand_format(void) { return VPT_OPERAND_NAME; }
const char* vpt_last_error(void) { return g_err; }

long vpt_conv3x3_packed_elems(int Cout, int Cin) { return (long)((Cout + 127) / 128) * (Cin / 32) * 9 * 128 * 32; }
long vpt_conv3x3_table_floats(int Cout) { return 9L * ((Cout + 127) / 128) * 128; }
long vpt_linear_packed_elems(int N, int K) { return (long)((N + 127) / 128) * 128 * K; }

int vpt_pack_conv3x3(const float* weight, const float* gain, const float* bias, void* wpk, float* edge_sa, float* edge_sg,
                     int Cout, int Cin, void* stream) {
  if ((edge_sa == nullptr) != (edge_sg == nullptr)) return fail(-1, "vpt_pack_conv3x3: give both edge tables or neither");
  if (edge_sa && !bias) return fail(-1, "vpt_pack_conv3x3: the edge tables need the GroupNorm bias");
  VptPackConvArgs a = {};
  a.weight = weight; a.gain = gain; a.bias = bias; a.wpk = (vpt_op16*)wpk; a.edge_sa = edge_sa; a.edge_sg = edge_sg;
  a.Cout = Cout; a.Cin = Cin; a.NT = (Cout + 127) / 128;
  CHECK_LAUNCH(vpt_pack_conv3x3_launch(&a, (hipStream_t)stream), "vpt_pack_conv3x3");
}

int vpt_pack_linear(const float* weight, void* wpk, int N, int K, int transposed, int ldw, int src_rows, void* stream) {
  CHECK_LAUNCH(vpt_pack_linear_launch(weight, wpk, N, K, transposed, ldw, src_rows, (hipStream_t)stream), "vpt_pack_linear");
}

long vpt_conv_first_packed_elems(int Cout) { return (long)((Cout + 127) / 128) * 4 * 2 * 64 * 8; }
long vpt_conv3d_t5_packed_elems(int O) { return (long)((O + 127) / 128) * 4 * 64 * 8; }

int vpt_pack_conv_first(const float* weight, const float* bias, void* wfrag, int Cout, void* stream) {
  if (!weight || !bias || !wfrag) return fail(-1, "vpt_pack_conv_first: null pointer");
  CHECK_LAUNCH(vpt_pack_conv_first_launch(weight, bias, wfrag, Cout, (hipStream_t)stream), "vpt_pack_conv_first");
}

int vpt_pack_conv3d_t5(const float* weight, const float* bias, void* wfrag, float* bias_padded, int O, void* stream) {
  if (!weight || !bias || !wfrag || !bias_padded) return fail(-1, "vpt_pack_conv3d_t5: null pointer");
  CHECK_LAUNCH(vpt_pack_conv3d_t5_launch(weight, bias, wfrag, bias_padded, O, (hipStream_t)stream), "vpt_pack_conv3d_t5");
}

int vpt_chw_to_blocked(const float* src, float* dst, int64_t rows, int C, int H, int W, void* stream) {
  if (!src || !dst || src == dst) return fail(-1, "vpt_chw_to_blocked: null or aliased pointers");
  CHECK_LAUNCH(vpt_chw_to_blocked_launch(src, dst, (long)rows, C, H, W, (hipStream_t)stream), "vpt_chw_to_blocked");
}

/* Bytes of caller-owned scratch an entry point needs (every buffer is the caller's: the library never allocates). */
int64_t vpt_workspace_bytes(int op, int frames, int H, int W, int Cin, int Cout) {
  switch (op) {
    case VPT_WS_CONV3X3_WGRAD: return 4 * (int64_t)vpt_conv3x3_wgrad_scratch_floats(frames, Cin, Cout);
    case VPT_WS_CONV_BACKWARD_PREPARE: return 4 * (int64_t)vpt_conv_bwd_prep_scratch_floats(frames, Cout);
    case VPT_WS_LAYERNORM_BACKWARD: return 4 * (int64_t)((frames + 31) / 32) * 4 * 2 * H;                   /* (M, D): one row per wave of a 32-row workgroup */
    case VPT_WS_COLUMN_SUM: return 4 * (int64_t)vpt_colsum_partial_floats(frames, H);                       /* (M, N) */
    case VPT_WS_ATTENTION_BACKWARD_DKV: return 4 * (int64_t)vpt_attn_bwd_dkv_floats(frames, H, W);          /* (B, t, hid) */
    case VPT_WS_ATTENTION_BACKWARD_DBND: return 4 * (int64_t)vpt_attn_bwd_dbnd_floats(frames, H, W, Cin);   /* (B, t, heads, maxlen) */
    case VPT_WS_FRAME_AFFINE_BACKWARD: return 4 * (int64_t)vpt_affine_bwd_partial_floats(frames, Cout / 32, H, W, Cin);   /* (frames, HW, per_element, pass, C) */
    case VPT_WS_CONV_FIRST_BACKWARD: return 4 * (int64_t)vpt_conv_first_bwd_partial_floats(frames, H, W, Cout);
    case VPT_WS_LINEAR_SPLITK: return 4 * (int64_t)frames * H * (int64_t)W;   /* splitk (= frames) x M (= H) x N (= W) fp32 partial slices */
    default: return -1;
  }
}

int vpt_conv_first_forward(const uint8_t* img, const void* wfrag, void* y, double* stats_out, const float* out_gain, double* chs_out,
                           int frames, int H, int W, int Cout, void* stream) {
  if (chs_out && Cout > 128) return fail(-1, "vpt_conv_first_forward: chs_out needs Cout <= 128 (use vpt_channel_stats)");
  VptConvFirstArgs a = {};
  a.img = img; a.wfrag = (const vpt_op16*)wfrag; a.y = (vpt_op16*)y; a.stats_out = stats_out; a.out_gain = out_gain; a.chs_out = chs_out;
  a.frames = frames; a.H = H; a.W = W; a.Cout = Cout; a.NT = (Cout + 127) / 128;
  CHECK_LAUNCH(vpt_conv_first_launch(&a, (hipStream_t)stream), "vpt_conv_first_forward");
}

int vpt_conv3d_t5_forward(const uint8_t* img, const void* wfrag, const float* bias, void* y, double* stats_out,
                          int frames, int T, int H, int W, int Cout, void* stream) {
  VptConv3dArgs a = {};
  a.img = img; a.wfrag = (const vpt_op16*)wfrag; a.bias = bias; a.y = (vpt_op16*)y; a.stats_out = stats_out;
  a.frames = frames; a.T = T; a.H = H; a.W = W; a.Cout = Cout; a.NT = (Cout + 127) / 128;
  CHECK_LAUNCH(vpt_conv3d_launch(&a, (hipStream_t)stream), "vpt_conv3d_t5_forward");
}

int vpt_conv3x3_forward(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg,
                        const double* stats_in, const void* res, void* y, double* stats_out,
                        int frames, int H, int W, int Cin, int Cout, void* stream) {
  return vpt_conv3x3_forward_tiled(x, wpk, edge_sa, edge_sg, stats_in, res, y, stats_out, frames, H, W, Cin, Cout, 1, stream);
}

int vpt_conv3x3_forward_tiled(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg,
                              const double* stats_in, const void* res, void* y, double* stats_out,
                              int frames, int H, int W, int Cin, int Cout, int tiling, void* stream) {
  if (!stats_in) return fail(-1, "vpt_conv3x3_forward: stats_in is required");
  if (tiling < 1 || tiling > 3) return fail(-1, "vpt_conv3x3_forward_tiled: tiling must be 1 (throughput), 2 (latency) or 3 (throughput, 32-row tiles)");
  VptConv3x3Args a = {};
  a.gate_stats = nullptr; a.gate_u = nullptr; a.inv_count_gate = 0.0; a.pool_mask = nullptr;
  a.tiling = tiling;
  a.x = (const vpt_op16*)x; a.wpk = (const vpt_op16*)wpk; a.edge_sa = edge_sa; a.edge_sg = edge_sg;
  a.stats_in = stats_in; a.res = (const vpt_op16*)res; a.y = (vpt_op16*)y; a.stats_out = stats_out;
  a.frames = frames; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.NT = (Cout + 127) / 128; a.CoutPad = a.NT * 128;
  a.inv_count_in = 1.0 / ((double)Cin * H * W);
  a.bwd = 0; a.xin = nullptr; a.coef = nullptr; a.pool = 0; a.seam_r = nullptr; a.seam_c = nullptr; a.out_gain = nullptr; a.chs_out = nullptr;
  a.kk_frame = a.rs_frame = a.res_scale = a.res_bias = nullptr;
  CHECK_LAUNCH(vpt_conv3x3_launch(&a, (hipStream_t)stream), "vpt_conv3x3_forward");
}

int vpt_conv3x3_forward_folded(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg, const double* stats_in,
                               const float* kk_frame, const float* rs_frame, const void* res, const float* res_scale, const float* res_bias,
                               void* y, double* stats_out, int frames, int H, int W, int Cin, int Cout, void* stream) {
  if ((kk_frame == nullptr) != (rs_frame == nullptr)) return fail(-1, "vpt_conv3x3_forward_folded: kk_frame and rs_frame go together");
  if (!kk_frame && !stats_in) return fail(-1, "vpt_conv3x3_forward_folded: stats_in is required unless kk_frame replaces it");
  if (res_bias && (!res || !res_scale)) return fail(-1, "vpt_conv3x3_forward_folded: res_bias needs res and res_scale");
  if (!edge_sg || (!kk_frame && !edge_sa)) return fail(-1, "vpt_conv3x3_forward_folded: edge tables missing");
  VptConv3x3Args a = {};
  a.gate_stats = nullptr; a.gate_u = nullptr; a.inv_count_gate = 0.0; a.pool_mask = nullptr;
  a.tiling = 1;
  a.x = (const vpt_op16*)x; a.wpk = (const vpt_op16*)wpk; a.edge_sa = edge_sa; a.edge_sg = edge_sg;
  a.stats_in = stats_in; a.res = (const vpt_op16*)res; a.y = (vpt_op16*)y; a.stats_out = stats_out;
  a.frames = frames; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.NT = (Cout + 127) / 128; a.CoutPad = a.NT * 128;
  a.inv_count_in = 1.0 / ((double)Cin * H * W);
  a.bwd = 0; a.xin = nullptr; a.coef = nullptr; a.pool = 0; a.seam_r = nullptr; a.seam_c = nullptr; a.out_gain = nullptr; a.chs_out = nullptr;
  a.kk_frame = kk_frame; a.rs_frame = rs_frame; a.res_scale = res_bias ? res_scale : nullptr; a.res_bias = res_bias;
  CHECK_LAUNCH(vpt_conv3x3_launch(&a, (hipStream_t)stream), "vpt_conv3x3_forward_folded");
}

int vpt_channel_stats(const void* x, double* chs, int frames, int C, int HW, void* stream) {
  if (!x || !chs || (C & 31)) return fail(-1, "vpt_channel_stats: null pointer or C not a multiple of 32");
  VptChannelStatsArgs a = {};
  a.x = (const vpt_op16*)x; a.chs = chs; a.frames = frames; a.CB = C / 32; a.HW = HW; a.split = 1;
  CHECK_LAUNCH(vpt_channel_stats_launch(&a, (hipStream_t)stream), "vpt_channel_stats");
}

int vpt_nfold_coef(const double* tot, const double* chs, const float* gain, const float* bias, const float* sa, const float* sg,
                   const float* tb, const float* tg, float* kk_frame, float* rs_frame, float* res_scale, float* res_bias,
                   int frames, int C, int HW, int Cout, void* stream) {
  if (!tot || !chs || !gain || !bias || !sa || !sg || !tb || !tg || !kk_frame || !rs_frame || !res_scale || !res_bias)
    return fail(-1, "vpt_nfold_coef: null pointer");
  VptNfoldCoefArgs a = {};
  a.tot = tot; a.chs = chs; a.gain = gain; a.bias = bias; a.sa = sa; a.sg = sg; a.tb = tb; a.tg = tg;
  a.kk_frame = kk_frame; a.rs_frame = rs_frame; a.res_scale = res_scale; a.res_bias = res_bias;
  a.frames = frames; a.C = C; a.HW = HW; a.CoutPad = ((Cout + 127) / 128) * 128;
  CHECK_LAUNCH(vpt_nfold_coef_launch(&a, (hipStream_t)stream), "vpt_nfold_coef");
}

int64_t vpt_conv3x3_pool_seam_elems(int frames, int H, int W, int Cout) { return (int64_t)frames * Cout * ((int64_t)(H / 16) * W + (int64_t)(W / 16) * H); }

static int conv3x3_pool_forward_impl(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg, const double* stats_in,
                                     void* pooled, void* pool_mask, void* seam_scratch, double* stats_out, const float* out_gain, double* chs_out, int frames, int H, int W,
                                     int Cin, int Cout, int phases, void* stream) {
  if (!stats_in || !pooled || !seam_scratch) return fail(-1, "vpt_conv3x3_pool_forward: stats_in, pooled and seam_scratch are required");
  if (phases < 1 || phases > 3) return fail(-1, "vpt_conv3x3_pool_forward: phases = 1 (tiles), 2 (seams) or 3 (both)");
  if ((H & 15) || (W & 15) || (Cout & 31)) return fail(-1, "vpt_conv3x3_pool_forward: H, W multiples of 16, Cout of 32");
  VptConv3x3Args a = {};
  a.gate_stats = nullptr; a.gate_u = nullptr; a.inv_count_gate = 0.0; a.pool_mask = nullptr;
  a.tiling = 1;
  a.x = (const vpt_op16*)x; a.wpk = (const vpt_op16*)wpk; a.edge_sa = edge_sa; a.edge_sg = edge_sg;
  a.stats_in = stats_in; a.res = nullptr; a.y = (vpt_op16*)pooled; a.stats_out = stats_out;
  a.frames = frames; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.NT = (Cout + 127) / 128; a.CoutPad = a.NT * 128;
  a.inv_count_in = 1.0 / ((double)Cin * H * W);
  a.bwd = 0; a.xin = nullptr; a.coef = nullptr; a.pool = 1; a.out_gain = out_gain; a.chs_out = chs_out; a.pool_mask = (vpt_op16*)pool_mask;
  a.kk_frame = a.rs_frame = a.res_scale = a.res_bias = nullptr;
  a.seam_r = (vpt_op16*)seam_scratch;
  a.seam_c = a.seam_r + (size_t)frames * Cout * (H / 16) * W;
  if (phases & 1) {
    int rc = vpt_conv3x3_launch(&a, (hipStream_t)stream);
    if (rc != 0) return fail(rc, "vpt_conv3x3_pool_forward (convolution)");
  }
  if (!(phases & 2)) return 0;
  VptPoolSeamArgs p = {};
  p.y = a.y; p.seam_r = a.seam_r; p.seam_c = a.seam_c; p.stats_out = stats_out; p.gain = out_gain; p.chs_out = chs_out; p.frames = frames; p.CB = Cout / 32; p.H = H; p.W = W;
  p.mask = (vpt_op16*)pool_mask;
  CHECK_LAUNCH(vpt_pool_seam_launch(&p, (hipStream_t)stream), "vpt_conv3x3_pool_forward (seams)");
}

int vpt_conv3x3_pool_forward(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg, const double* stats_in,
                             void* pooled, void* seam_scratch, double* stats_out, const float* out_gain, double* chs_out, int frames, int H, int W,
                             int Cin, int Cout, int phases, void* stream) {
  return conv3x3_pool_forward_impl(x, wpk, edge_sa, edge_sg, stats_in, pooled, nullptr, seam_scratch, stats_out, out_gain, chs_out, frames, H, W, Cin, Cout, phases, stream);
}

int vpt_conv3x3_pool_argmax_forward(const void* x, const void* wpk, const float* edge_sa, const float* edge_sg, const double* stats_in,
                                    void* pooled, void* pool_mask, void* seam_scratch, double* stats_out, int frames, int H, int W, int Cin, int Cout, int phases, void* stream) {
  if (!pool_mask) return fail(-1, "vpt_conv3x3_pool_argmax_forward: pool_mask is required");
  return conv3x3_pool_forward_impl(x, wpk, edge_sa, edge_sg, stats_in, pooled, pool_mask, seam_scratch, stats_out, nullptr, nullptr, frames, H, W, Cin, Cout, phases, stream);
}

int vpt_conv3x3_dgrad(const void* dacc, const void* wpk_t, const void* skip, const void* xin, const float* coef, void* dx,
                      int frames, int H, int W, int Cout, int Cin, void* stream) {
  VptConv3x3Args a = {};
  a.gate_stats = nullptr; a.gate_u = nullptr; a.inv_count_gate = 0.0; a.pool_mask = nullptr;
  a.x = (const vpt_op16*)dacc; a.wpk = (const vpt_op16*)wpk_t; a.edge_sa = nullptr; a.edge_sg = nullptr;
  a.stats_in = nullptr; a.res = (const vpt_op16*)skip; a.y = (vpt_op16*)dx; a.stats_out = nullptr;
  a.frames = frames; a.H = H; a.W = W; a.Cin = Cout; a.Cout = Cin;   // roles swap in the transposed convolution
  a.NT = (Cin + 127) / 128; a.CoutPad = a.NT * 128; a.inv_count_in = 1.0;
  a.bwd = 1; a.xin = (const vpt_op16*)xin; a.coef = coef; a.tiling = 1; a.pool = 0; a.seam_r = nullptr; a.seam_c = nullptr; a.out_gain = nullptr; a.chs_out = nullptr;
  a.kk_frame = a.rs_frame = a.res_scale = a.res_bias = nullptr;
  CHECK_LAUNCH(vpt_conv3x3_launch(&a, (hipStream_t)stream), "vpt_conv3x3_dgrad");
}

int vpt_conv3x3_dgrad_gated(const void* dacc, const void* wpk_t, const void* xin, const float* coef, const double* gate_stats, int gate_cin,
                            void* dacc_out, double* gate_u, int frames, int H, int W, int Cout, int Cin, void* stream) {
  if (!gate_stats || !gate_u || gate_cin <= 0) return fail(-1, "vpt_conv3x3_dgrad_gated: gate_stats, gate_u and gate_cin are required");
  VptConv3x3Args a = {};
  a.x = (const vpt_op16*)dacc; a.wpk = (const vpt_op16*)wpk_t; a.edge_sa = nullptr; a.edge_sg = nullptr;
  a.stats_in = nullptr; a.res = nullptr; a.y = (vpt_op16*)dacc_out; a.stats_out = nullptr;
  a.frames = frames; a.H = H; a.W = W; a.Cin = Cout; a.Cout = Cin;   // roles swap in the transposed convolution
  a.NT = (Cin + 127) / 128; a.CoutPad = a.NT * 128; a.inv_count_in = 1.0;
  a.bwd = 1; a.xin = (const vpt_op16*)xin; a.coef = coef; a.tiling = 1; a.pool = 0; a.seam_r = nullptr; a.seam_c = nullptr; a.out_gain = nullptr; a.chs_out = nullptr;
  a.kk_frame = a.rs_frame = a.res_scale = a.res_bias = nullptr;
  a.gate_stats = gate_stats; a.gate_u = gate_u; a.inv_count_gate = 1.0 / ((double)gate_cin * H * W);
  CHECK_LAUNCH(vpt_conv3x3_launch(&a, (hipStream_t)stream), "vpt_conv3x3_dgrad_gated");
}

int vpt_conv_backward_reduce(const void* dacc, const double* gate_u, const double* stats_in, const float* edge_sa, const float* edge_sg,
                             double* t12, float* coef, float* d_sa, float* d_sg, float* scratch, int frames, int H, int W, int Cin, int Cout, void* stream) {
  if (Cout & 31) return fail(-1, "vpt_conv_backward_reduce: Cout must be a multiple of 32");
  if (!dacc || !gate_u) return fail(-1, "vpt_conv_backward_reduce: dacc and gate_u are required");
  VptConvBwdPrepArgs a = {};
  a.dpooled = nullptr; a.argmax = nullptr; a.sbuf = scratch; a.wshift = 0; a.coef = coef;
  a.dy = (const vpt_op16*)dacc; a.y = nullptr; a.res = nullptr; a.stats_in = stats_in;
  a.edge_sa = edge_sa; a.edge_sg = edge_sg; a.dacc = nullptr; a.t12 = t12; a.d_sa = d_sa; a.d_sg = d_sg;
  a.frames = frames; a.CB = Cout / 32; a.H = H; a.W = W; a.CoutPad = ((Cout + 127) / 128) * 128;
  a.inv_count_in = 1.0 / ((double)Cin * H * W);
  a.gate_u = gate_u; a.pooled = nullptr; a.pool_mask = nullptr;
  CHECK_LAUNCH(vpt_conv_bwd_prep_launch(&a, (hipStream_t)stream), "vpt_conv_backward_reduce");
}

int vpt_conv_backward_prepare_pooled(const void* dpooled, const void* pooled, const void* pool_mask, const double* stats_in, const float* edge_sa, const float* edge_sg,
                                     void* dacc, double* t12, float* coef, float* d_sa, float* d_sg, float* scratch,
                                     const float* n_gain, const double* pool_stats, const double* pool_ab, int frames, int H, int W, int Cin, int Cout, void* stream) {
  if (Cout & 31) return fail(-1, "vpt_conv_backward_prepare_pooled: Cout must be a multiple of 32");
  if (!dpooled || !pooled || !pool_mask || !dacc) return fail(-1, "vpt_conv_backward_prepare_pooled: dpooled, pooled, pool_mask and dacc are required");
  VptConvBwdPrepArgs a = {};
  a.dpooled = (const vpt_op16*)dpooled; a.argmax = nullptr; a.sbuf = scratch; a.wshift = 0; a.coef = coef;
  a.dy = nullptr; a.y = nullptr; a.res = nullptr; a.stats_in = stats_in;
  a.edge_sa = edge_sa; a.edge_sg = edge_sg; a.dacc = (vpt_op16*)dacc; a.t12 = t12; a.d_sa = d_sa; a.d_sg = d_sg;
  a.frames = frames; a.CB = Cout / 32; a.H = H; a.W = W; a.CoutPad = ((Cout + 127) / 128) * 128;
  a.inv_count_in = 1.0 / ((double)Cin * H * W);
  a.gate_u = nullptr; a.pooled = (const vpt_op16*)pooled; a.pool_mask = (const vpt_op16*)pool_mask;
  a.n_gain = n_gain; a.pool_stats = pool_stats; a.pool_ab = pool_ab; a.inv_count_pool = 1.0 / ((double)Cout * (H / 2) * (W / 2));
  CHECK_LAUNCH(vpt_conv_bwd_prep_launch(&a, (hipStream_t)stream), "vpt_conv_backward_prepare_pooled");
}

int vpt_conv_backward_prepare(const void* dy, const void* dpooled, const uint8_t* argmax, const void* y, const void* res,
                              const double* stats_in, const float* edge_sa, const float* edge_sg, void* dacc, double* t12,
                              float* coef, float* d_sa, float* d_sg, float* scratch, int frames, int H, int W, int Cin, int Cout, void* stream) {
  if (Cout & 31) return fail(-1, "vpt_conv_backward_prepare: Cout must be a multiple of 32");
  VptConvBwdPrepArgs a = {};
  a.dpooled = (const vpt_op16*)dpooled; a.argmax = argmax; a.sbuf = scratch; a.wshift = 0; a.coef = coef;
  a.dy = (const vpt_op16*)dy; a.y = (const vpt_op16*)y; a.res = (const vpt_op16*)res; a.stats_in = stats_in;
  a.edge_sa = edge_sa; a.edge_sg = edge_sg; a.dacc = (vpt_op16*)dacc; a.t12 = t12; a.d_sa = d_sa; a.d_sg = d_sg;
  a.frames = frames; a.CB = Cout / 32; a.H = H; a.W = W; a.CoutPad = ((Cout + 127) / 128) * 128;
  a.inv_count_in = 1.0 / ((double)Cin * H * W);
  a.gate_u = nullptr; a.pooled = nullptr; a.pool_mask = nullptr;
  CHECK_LAUNCH(vpt_conv_bwd_prep_launch(&a, (hipStream_t)stream), "vpt_conv_backward_prepare");
}

int vpt_conv_first_backward(const uint8_t* img, const void* wfrag, const void* dpooled, float* dw, float* db, float* partials,
                            int frames, int H, int W, int Cout, void* stream) {
  if (!partials) return fail(-1, "vpt_conv_first_backward: partials workspace is required");
  VptConvFirstBwdArgs a = {};
  a.img = img; a.wfrag = (const vpt_op16*)wfrag; a.dpooled = (const vpt_op16*)dpooled; a.dw = dw; a.db = db; a.partials = partials;
  a.frames = frames; a.H = H; a.W = W; a.Cout = Cout;
  CHECK_LAUNCH(vpt_conv_first_bwd_launch(&a, (hipStream_t)stream), "vpt_conv_first_backward");
}

int vpt_conv_first_backward_nfold(const uint8_t* img, const void* wfrag, const void* g, const float* n_gain, const double* pool_stats, const double* pool_ab,
                                  float* dw, float* db, float* partials, int frames, int H, int W, int Cout, void* stream) {
  if (!partials || !n_gain || !pool_stats || !pool_ab) return fail(-1, "vpt_conv_first_backward_nfold: partials, n_gain, pool_stats and pool_ab are required");
  VptConvFirstBwdArgs a = {};
  a.img = img; a.wfrag = (const vpt_op16*)wfrag; a.dpooled = (const vpt_op16*)g; a.dw = dw; a.db = db; a.partials = partials;
  a.n_gain = n_gain; a.pool_stats = pool_stats; a.pool_ab = pool_ab; a.inv_count_pool = 1.0 / ((double)Cout * (H / 2) * (W / 2));
  a.frames = frames; a.H = H; a.W = W; a.Cout = Cout;
  CHECK_LAUNCH(vpt_conv_first_bwd_launch(&a, (hipStream_t)stream), "vpt_conv_first_backward_nfold");
}

long vpt_conv3x3_wgrad_scratch_floats(int frames, int Cin, int Cout) {
  return (long)vpt_conv_wgrad_groups(frames, Cin, Cout) * Cout * 9 * Cin;
}

int vpt_conv3x3_wgrad(const void* dacc, const void* x, float* dw, float* scratch, int frames, int H, int W, int Cin, int Cout, void* stream) {
  VptConvWgradArgs a = {};
  a.dacc = (const vpt_op16*)dacc; a.x = (const vpt_op16*)x; a.dw = dw; a.partial = scratch;
  a.frames = frames; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.OT = 0; a.frames_per_wg = 0;
  CHECK_LAUNCH(vpt_conv_wgrad_launch(&a, (hipStream_t)stream), "vpt_conv3x3_wgrad");
}

int vpt_maxpool_backward(const void* pre, const void* pooled, const void* dpooled, void* dpre,
                         int frames, int C, int H, int W, void* stream) {
  VptPoolBwdArgs a = {};
  a.pre = (const vpt_op16*)pre; a.pooled = (const vpt_op16*)pooled; a.dpooled = (const vpt_op16*)dpooled;
  a.dpre = (vpt_op16*)dpre; a.frames = frames; a.CB = C / 32; a.H = H; a.W = W;
  CHECK_LAUNCH(vpt_pool_bwd_launch(&a, (hipStream_t)stream), "vpt_maxpool_backward");
}

int vpt_frame_affine_backward(const void* x, const void* dy, const void* dx_add, void* dx, const float* gain,
                              const double* stats_in, double* ab, float* dgain, float* dbias, float* partials,
                              int frames, int C, int HW, int per_element, int pass, void* stream) {
  VptAffineBwdArgs a = {};
  a.partials = partials;
  a.x = (const vpt_op16*)x; a.dy = (const vpt_op16*)dy; a.dx_add = (const vpt_op16*)dx_add; a.dx = (vpt_op16*)dx;
  a.gain = gain; a.stats_in = stats_in; a.ab = ab; a.dgain = dgain; a.dbias = dbias;
  a.frames = frames; a.CB = C / 32; a.HW = HW; a.per_element = per_element; a.inv_count = 1.0 / ((double)C * HW);
  CHECK_LAUNCH(vpt_affine_bwd_launch(&a, pass, (hipStream_t)stream), "vpt_frame_affine_backward");
}

int vpt_maxpool_forward(const void* x, void* y, double* stats_out, uint8_t* argmax, int frames, int C, int H, int W, void* stream) {
  if (C & 31) return fail(-1, "vpt_maxpool_forward: C must be a multiple of 32");
  VptPoolArgs a = {};
  a.x = (const vpt_op16*)x; a.y = (vpt_op16*)y; a.stats_out = stats_out; a.argmax = argmax;
  a.frames = frames; a.CB = C / 32; a.H = H; a.W = W;
  CHECK_LAUNCH(vpt_pool_launch(&a, (hipStream_t)stream), "vpt_maxpool_forward");
}

int vpt_frame_affine_forward(const void* x, void* y, const float* gain, const float* bias,
                             const double* stats_in, double* stats_out,
                             int frames, int C, int HW, int per_element, void* stream) {
  if (C & 31) return fail(-1, "vpt_frame_affine_forward: C must be a multiple of 32");
  VptAffineArgs a = {};
  a.x = (const vpt_op16*)x; a.y = (vpt_op16*)y; a.gain = gain; a.bias = bias;
  a.stats_in = stats_in; a.stats_out = stats_out;
  a.frames = frames; a.CB = C / 32; a.HW = HW; a.per_element = per_element;
  a.inv_count = 1.0 / ((double)C * HW);
  CHECK_LAUNCH(vpt_affine_launch(&a, (hipStream_t)stream), "vpt_frame_affine_forward");
}

int vpt_linear_forward(const void* A, const void* wpk, const float* bias, const float* res,
                       float* out_f32, void* out_bf16, int M, int N, int K,
                       int lda, int ldr, int ldc, int ldcb, int relu, int splitk, const void* mask, int ldm,
                       void* stream) {
  return vpt_linear_forward_tiled(A, wpk, bias, res, out_f32, out_bf16, M, N, K, lda, ldr, ldc, ldcb, relu, splitk, mask, ldm, 0, stream);
}

int vpt_linear_forward_tiled(const void* A, const void* wpk, const float* bias, const float* res,
                             float* out_f32, void* out_bf16, int M, int N, int K,
                             int lda, int ldr, int ldc, int ldcb, int relu, int splitk, const void* mask, int ldm,
                             int tiling, void* stream) {
  if (tiling < 0 || tiling > 4 || tiling == 3) return fail(-1, "vpt_linear_forward_tiled: tiling must be 0 (by M), 1 (MFMA GEMM), 2 (weight-streaming, M <= 8) or 4 (MFMA GEMM, 256 x 256 tiles where the grid fills the chip)");
  VptGemmArgs a{};
  a.tiling = tiling;
  a.mask = (const vpt_op16*)mask; a.ldm = ldm;
  a.A = (const vpt_op16*)A; a.wpk = (const vpt_op16*)wpk; a.bias = bias; a.res = res;
  a.out_f32 = out_f32; a.out_bf16 = (vpt_op16*)out_bf16;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldr = ldr; a.ldc = ldc; a.ldcb = ldcb;
  a.relu = relu; a.splitk = splitk < 1 ? 1 : splitk; a.atomic_out = a.splitk > 1;
  CHECK_LAUNCH(vpt_gemm_launch(&a, (hipStream_t)stream), "vpt_linear_forward");
}

int vpt_layernorm_linear_forward(const float* x, const float* ln_gain, const float* ln_bias, int relu_in, float* ln_out_f32,
                                 const void* wpk, const float* bias, const float* res, float* out_f32, void* out_bf16,
                                 int M, int N, int K, int ldr, int ldc, int ldcb, int relu, void* stream) {
  if (M <= 0 || M > 8 || K > 3072 || !x || !ln_gain || !ln_bias)
    return fail(-1, "vpt_layernorm_linear_forward: M <= 8 rows and K <= 3072 only (use vpt_layernorm_forward + vpt_linear_forward)");
  VptGemmArgs a{};
  a.ln_x = x; a.ln_gain = ln_gain; a.ln_bias = ln_bias; a.ln_relu_in = relu_in; a.ln_out_f32 = ln_out_f32;
  a.wpk = (const vpt_op16*)wpk; a.bias = bias; a.res = res; a.out_f32 = out_f32; a.out_bf16 = (vpt_op16*)out_bf16;
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldr = ldr; a.ldc = ldc; a.ldcb = ldcb; a.relu = relu; a.splitk = 1; a.atomic_out = 0;
  CHECK_LAUNCH(vpt_gemm_launch(&a, (hipStream_t)stream), "vpt_layernorm_linear_forward");   // M <= 8: dispatches to vpt_gemv_launch
}

int vpt_linear_wgrad(const void* dy, const void* x, float* dw, int M, int N, int K, int ldy, int ldx, int ldw, int accumulate, void* stream) {
  VptGemmTnArgs a = {};
  a.A = (const vpt_op16*)dy; a.B = (const vpt_op16*)x; a.C = dw; a.M = M; a.N1 = N; a.N2 = K; a.lda = ldy; a.ldb = ldx; a.ldc = ldw;
  a.accumulate = accumulate;
  CHECK_LAUNCH(vpt_gemm_tn_launch(&a, (hipStream_t)stream), "vpt_linear_wgrad");
}

int vpt_dense_fold_epilogue(const float* part, int splitk, const double* stats, int count, const float* sg, const float* sb, float* out,
                            int M, int N, void* stream) {
  if (count <= 0) return fail(-1, "vpt_dense_fold_epilogue: count = elements per frame of the normalised tensor");
  CHECK_LAUNCH(vpt_dense_fold_epilogue_launch(part, splitk, stats, 1.0 / (double)count, sg, sb, out, M, N, (hipStream_t)stream), "vpt_dense_fold_epilogue");
}

int vpt_linear_splitk_epilogue(const float* part, int splitk, const float* bias, const float* res, float* out_f32, void* out_bf16,
                               int M, int N, int ldr, int ldc, int ldcb, int relu, const void* mask, int ldm, void* stream) {
  VptGemmArgs a{};
  a.A = nullptr; a.wpk = nullptr; a.bias = bias; a.res = res; a.out_f32 = out_f32; a.out_bf16 = (vpt_op16*)out_bf16;
  a.M = M; a.N = N; a.K = 0; a.lda = 0; a.ldr = ldr; a.ldc = ldc; a.ldcb = ldcb; a.relu = relu; a.splitk = splitk; a.atomic_out = 0;
  a.mask = (const vpt_op16*)mask; a.ldm = ldm;
  CHECK_LAUNCH(vpt_splitk_epilogue_launch(part, splitk, &a, (hipStream_t)stream), "vpt_linear_splitk_epilogue");
}

int vpt_layernorm_forward(const float* x, const float* gain, const float* bias, float* out_f32, void* out_bf16,
                          int M, int D, int relu_in, void* stream) {
  VptLayerNormArgs a = {};
  a.x = x; a.gain = gain; a.bias = bias; a.out_f32 = out_f32; a.out_bf16 = (vpt_op16*)out_bf16;
  a.M = M; a.D = D; a.relu_in = relu_in;
  CHECK_LAUNCH(vpt_layernorm_launch(&a, (hipStream_t)stream), "vpt_layernorm_forward");
}

int vpt_masked_attention_forward(const float* qkvr, const float* kmem, const float* vmem, const uint8_t* memvalid,
                                 const float* b_nd, void* out, int B, int t, int heads, int hid, int ld,
                                 int maxlen, int causal, void* stream) {
  VptAttnArgs a = {};
  a.qkvr = qkvr; a.kmem = kmem; a.vmem = vmem; a.memvalid = memvalid; a.b_nd = b_nd; a.out = (vpt_op16*)out;
  a.B = B; a.t = t; a.heads = heads; a.hid = hid; a.ld = ld; a.maxlen = maxlen; a.causal = causal;
  CHECK_LAUNCH(vpt_attn_launch(&a, (hipStream_t)stream), "vpt_masked_attention_forward");
}

int vpt_masked_attention_step(const float* qkvr, const float* kmem, const float* vmem, const uint8_t* state_mask, const uint8_t* first,
                              const float* b_nd, void* out, float* kout, float* vout, uint8_t* mask_out,
                              int B, int heads, int hid, int ld, int maxlen, void* stream) {
  VptAttnArgs a = {};
  a.qkvr = qkvr; a.kmem = kmem; a.vmem = vmem; a.memvalid = nullptr; a.b_nd = b_nd; a.out = (vpt_op16*)out;
  a.B = B; a.t = 1; a.heads = heads; a.hid = hid; a.ld = ld; a.maxlen = maxlen; a.causal = 1;
  CHECK_LAUNCH(vpt_attn_step_launch(&a, state_mask, first, mask_out, kout, vout, nullptr, (hipStream_t)stream), "vpt_masked_attention_step");
}

int vpt_masked_attention_step_inplace(const float* qkvr, float* kmem, float* vmem, uint8_t* state_mask, const uint8_t* first,
                                      const float* b_nd, void* out, int* done_counter, int B, int heads, int hid, int ld, int maxlen, void* stream) {
  if (!done_counter) return fail(-1, "vpt_masked_attention_step_inplace: done_counter ([B] ints, zero before the first launch) is required");
  VptAttnArgs a = {};
  a.qkvr = qkvr; a.kmem = kmem; a.vmem = vmem; a.memvalid = nullptr; a.b_nd = b_nd; a.out = (vpt_op16*)out;
  a.B = B; a.t = 1; a.heads = heads; a.hid = hid; a.ld = ld; a.maxlen = maxlen; a.causal = 1;
  CHECK_LAUNCH(vpt_attn_step_launch(&a, state_mask, first, state_mask, kmem, vmem, done_counter, (hipStream_t)stream), "vpt_masked_attention_step_inplace");
}

int vpt_act_epilogue(const int64_t* action_buttons, const int64_t* action_camera, const float* logp_buttons, const float* logp_camera,
                     const float* logits, int ld, int value_col, float scale, float shift, int64_t* keep, uint8_t* nan_flag, uint64_t* rng_state,
                     int B, void* stream) {
  CHECK_LAUNCH(vpt_act_epilogue_launch(action_buttons, action_camera, logp_buttons, logp_camera, logits, ld, value_col, scale, shift, keep, nan_flag,
                                       rng_state, B, (hipStream_t)stream), "vpt_act_epilogue");
}

int vpt_uniform_noise(const uint64_t* rng_state, uint32_t rng_stream, float* out, int M, int n, void* stream) {
  CHECK_LAUNCH(vpt_uniform_noise_launch(rng_state, rng_stream, out, M, n, (hipStream_t)stream), "vpt_uniform_noise");
}

int vpt_kv_memory_update(const float* qkvr, const float* kmem, const float* vmem, float* kout, float* vout,
                         int B, int t, int hid, int ld, int maxlen, void* stream) {
  VptKvUpdateArgs a = {};
  a.qkvr = qkvr; a.kmem = kmem; a.vmem = vmem; a.kout = kout; a.vout = vout;
  a.B = B; a.t = t; a.hid = hid; a.ld = ld; a.maxlen = maxlen;
  CHECK_LAUNCH(vpt_kv_update_launch(&a, (hipStream_t)stream), "vpt_kv_memory_update");
}

int vpt_log_softmax_forward(const float* logits, float* out, int M, int ld, int col0, int n, float temperature,
                            void* stream) {
  VptLogSoftmaxArgs a = {};
  a.logits = logits; a.out = out; a.M = M; a.ld = ld; a.col0 = col0; a.n = n; a.temperature = temperature;
  a.mask = nullptr; a.noise = nullptr; a.action = nullptr; a.action_logp = nullptr; a.rng_state = nullptr; a.rng_stream = 0;
  CHECK_LAUNCH(vpt_logsoftmax_launch(&a, (hipStream_t)stream), "vpt_log_softmax_forward");
}

int vpt_action_head_forward(const float* logits, const uint8_t* mask, const float* noise, const uint64_t* rng_state, uint32_t rng_stream,
                            float* out, int64_t* action, float* action_logp, int M, int ld, int col0, int n, float temperature, void* stream) {
  if (action_logp && !action) return fail(-1, "vpt_action_head_forward: action_logp needs action");
  if (noise && rng_state) return fail(-1, "vpt_action_head_forward: give the uniforms (noise) or the generator state (rng_state), not both");
  VptLogSoftmaxArgs a = {};
  a.logits = logits; a.out = out; a.M = M; a.ld = ld; a.col0 = col0; a.n = n; a.temperature = temperature;
  a.mask = mask; a.noise = noise; a.action = (long*)action; a.action_logp = action_logp; a.rng_state = rng_state; a.rng_stream = rng_stream;
  CHECK_LAUNCH(vpt_logsoftmax_launch(&a, (hipStream_t)stream), "vpt_action_head_forward");
}

int vpt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  if (step < 1) return fail(-1, "vpt_adam_step: step counts from 1");
  VptAdamArgs a = {};
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = (size_t)n; a.skip_flag = nullptr;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.grad_scale = grad_scale;
  a.step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  CHECK_LAUNCH(vpt_adam_launch(&a, (hipStream_t)stream), "vpt_adam_step");
}

int vpt_adam_step_multi(const void* table, int ntensors, int64_t total_blocks, int step, float lr, float beta1, float beta2,
                        float eps, float weight_decay, float grad_scale, const int32_t* skip_flag, void* stream) {
  if (step < 1) return fail(-1, "vpt_adam_step_multi: step counts from 1");
  if (!table && ntensors > 0) return fail(-1, "vpt_adam_step_multi: null table");
  VptAdamArgs a = {};
  a.p = nullptr; a.g = nullptr; a.m = nullptr; a.v = nullptr; a.n = 0; a.skip_flag = (const int*)skip_flag;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.grad_scale = grad_scale;
  a.step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  CHECK_LAUNCH(vpt_adam_multi_launch((const VptAdamTensor*)table, ntensors, (long)total_blocks, &a, (hipStream_t)stream), "vpt_adam_step_multi");
}

int vpt_grads_nonfinite_multi(const void* table, int ntensors, int64_t total_blocks, int32_t* flag, void* stream) {
  if ((!table && ntensors > 0) || !flag) return fail(-1, "vpt_grads_nonfinite_multi: null table / flag");
  CHECK_LAUNCH(vpt_grads_nonfinite_launch((const VptAdamTensor*)table, ntensors, (long)total_blocks, (int*)flag, (hipStream_t)stream), "vpt_grads_nonfinite_multi");
}

int vpt_bc_nll_backward(const float* lp_buttons, const float* lp_camera, const int64_t* act_buttons,
                        const int64_t* act_camera, void* dz, int M, int nb, int nc, int ldz, float scale, void* stream) {
  VptNllBwdArgs a = {};
  a.lp_buttons = lp_buttons; a.lp_camera = lp_camera; a.act_buttons = (const long*)act_buttons;
  a.act_camera = (const long*)act_camera; a.dz = (vpt_op16*)dz; a.M = M; a.nb = nb; a.nc = nc; a.ldz = ldz; a.scale = scale;
  CHECK_LAUNCH(vpt_nll_bwd_launch(&a, (hipStream_t)stream), "vpt_bc_nll_backward");
}

int vpt_heads_logprob_backward(const float* lp_buttons, const float* lp_camera, const float* g_buttons, const float* g_camera,
                               const float* g_value, const uint8_t* mask_buttons, const uint8_t* mask_camera, void* dz,
                               int M, int nb, int nc, int ldz, float temperature, float grad_scale, void* stream) {
  if (!(temperature > 0.f)) return fail(-1, "vpt_heads_logprob_backward: temperature must be positive");
  VptHeadsBwdArgs a = {};
  a.lp_buttons = lp_buttons; a.lp_camera = lp_camera; a.g_buttons = g_buttons; a.g_camera = g_camera; a.g_value = g_value;
  a.mask_buttons = mask_buttons; a.mask_camera = mask_camera;
  a.dz = (vpt_op16*)dz; a.M = M; a.nb = nb; a.nc = nc; a.ldz = ldz; a.inv_temp = 1.0f / temperature; a.grad_scale = grad_scale;
  CHECK_LAUNCH(vpt_heads_bwd_launch(&a, (hipStream_t)stream), "vpt_heads_logprob_backward");
}

int vpt_layernorm_backward(const float* x, const float* gain, const float* dy, const float* dx_add, float* dx,
                           float* dgain, float* dbias, float* partials, int M, int D, int relu_in, void* stream) {
  if (!partials) return fail(-1, "vpt_layernorm_backward: partials workspace is required");
  VptLnBwdArgs a = {};
  a.partials = partials;
  a.x = x; a.gain = gain; a.dy = dy; a.dx_add = dx_add; a.dx = dx; a.dgain = dgain; a.dbias = dbias;
  a.M = M; a.D = D; a.relu_in = relu_in;
  CHECK_LAUNCH(vpt_ln_bwd_launch(&a, (hipStream_t)stream), "vpt_layernorm_backward");
}

int vpt_gate_cast(const float* x, const void* mask, void* out, int M, int N, int ldx, int ldm, int ldo, void* stream) {
  VptGateCastArgs a = {};
  a.x = x; a.mask = (const vpt_op16*)mask; a.out = (vpt_op16*)out; a.M = M; a.N = N; a.ldx = ldx; a.ldm = ldm; a.ldo = ldo;
  CHECK_LAUNCH(vpt_gate_cast_launch(&a, (hipStream_t)stream), "vpt_gate_cast");
}

int vpt_column_sum(const void* x_bf16, float* out, float* partials, int M, int N, int ld, void* stream) {
  VptColsumArgs a = {};
  a.partials = partials;
  a.x = (const vpt_op16*)x_bf16; a.out = out; a.M = M; a.N = N; a.ld = ld;
  CHECK_LAUNCH(vpt_colsum_launch(&a, (hipStream_t)stream), "vpt_column_sum");
}

int vpt_masked_attention_backward(const float* qkvr, const float* kmem, const float* vmem, const uint8_t* memvalid,
                                  const float* b_nd, const float* dout, float* dqkvr, float* db_nd, float* dkv_slab, float* dbnd_slab,
                                  int B, int t, int heads, int hid, int ld, int maxlen, void* stream) {
  if (!dkv_slab || !dbnd_slab) return fail(-1, "vpt_masked_attention_backward: the dkv_slab / dbnd_slab workspaces are required");
  VptAttnBwdArgs a = {};
  a.dkv_slab = dkv_slab; a.dbnd_slab = dbnd_slab;
  a.qkvr = qkvr; a.kmem = kmem; a.vmem = vmem; a.memvalid = memvalid; a.b_nd = b_nd; a.dout = dout;
  a.dqkvr = dqkvr; a.db_nd = db_nd; a.B = B; a.t = t; a.heads = heads; a.hid = hid; a.ld = ld; a.maxlen = maxlen;
  CHECK_LAUNCH(vpt_attn_bwd_launch(&a, (hipStream_t)stream), "vpt_masked_attention_backward");
}

}  // extern "C"

/* ---- action codec (lib/actions.py, lib/action_mapping.py) ---- */
int vpt_camera_discretize(const double* xy, long* bins, long n, double maxval, double binsize, double mu, int mu_law, void* stream) {
  CHECK_LAUNCH(vpt_camera_codec_launch(0, xy, bins, n, maxval, binsize, mu, mu_law, (hipStream_t)stream), "vpt_camera_discretize");
}

int vpt_camera_undiscretize(const long* bins, double* xy, long n, double maxval, double binsize, double mu, int mu_law, void* stream) {
  CHECK_LAUNCH(vpt_camera_codec_launch(1, bins, xy, n, maxval, binsize, mu, mu_law, (hipStream_t)stream), "vpt_camera_undiscretize");
}

int vpt_action_from_factored(const long* buttons, const long* camera, long* joint_buttons, long* joint_camera, long n, int n_camera_bins, void* stream) {
  CHECK_LAUNCH(vpt_action_mapping_launch(0, buttons, camera, joint_buttons, joint_camera, n, n_camera_bins, (hipStream_t)stream), "vpt_action_from_factored");
}

int vpt_action_to_factored(const long* joint_buttons, const long* joint_camera, long* buttons, long* camera, long n, int n_camera_bins, void* stream) {
  CHECK_LAUNCH(vpt_action_mapping_launch(1, joint_buttons, joint_camera, buttons, camera, n, n_camera_bins, (hipStream_t)stream), "vpt_action_to_factored");
}

/* ---- clip data path (data_loader.py:34-46,113-122; agent.py:100-103) ---- */
int vpt_clip_frames(const uint8_t* src_bgr, int frames, int height, int width, const int32_t* cursor_state, const uint8_t* cursor_bgr,
                    const double* cursor_alpha, int cursor_h, int cursor_w, uint8_t* dst_rgb, int out_height, int out_width, void* stream) {
  if (!src_bgr || !dst_rgb) return fail(-1, "vpt_clip_frames");
  VptClipArgs a{};
  a.src = src_bgr; a.cursor = cursor_state; a.cursor_img = cursor_bgr; a.cursor_alpha = cursor_alpha; a.dst = dst_rgb;
  a.frames = frames; a.H = height; a.W = width; a.OH = out_height; a.OW = out_width; a.CH = cursor_h; a.CW = cursor_w;
  CHECK_LAUNCH(vpt_clip_launch(&a, (hipStream_t)stream), "vpt_clip_frames");
}

int vpt_debug_poison_lds(void* stream) {
  static unsigned long long optin_done = 0;
  const int bytes = 160 * 1024;
  if (!vpt_lds_optin((const void*)vpt_poison_lds_kernel, bytes, &optin_done)) return fail(-4, "vpt_debug_poison_lds: LDS opt-in");
  hipLaunchKernelGGL(vpt_poison_lds_kernel, dim3(1024), dim3(256), bytes, (hipStream_t)stream, bytes / 4, 0x7fc07fc0u);
  if (hipGetLastError() != hipSuccess) return fail(-3, "vpt_debug_poison_lds");
  return 0;
}
