// Clip data path on the device (gfx950): decoded video frame -> the 128 x 128 x 3 RGB uint8 frame the policy
// ingests, for a whole batch of frames in one launch.
//
// Replaces, per kept frame, data_loader.py:113-122:
//     composite_images_with_alpha(frame, cursor_image, cursor_alpha, cursor_x, cursor_y)      (:34-46, only while a GUI is open)
//     cv2.cvtColor(frame, COLOR_BGR2RGB);  np.clip(.., 0, 255)                                (:120-121; the clip is a no-op on uint8)
//     resize_image(frame, AGENT_RESOLUTION) = cv2.resize(frame, (128, 128), INTER_LINEAR)     (agent.py:100-103)
// Everything is integer / IEEE arithmetic, restated.  What is PINNED: the cursor blend is bit-identical to the live reference's
// composite_images_with_alpha (golden vectors, tests/golden/make_golden_clip.py).  What is NOT: the resize restates OpenCV's
// published algorithm and is bit-identical to oracle/clip_oracle.py, but neither has been checked against real cv2 output
// (opencv-python is absent from this image) -- "parity unpinned" for that one function.
//   * the cursor blend is numpy's  uint8(img * (1 - alpha) + cursor * alpha)  in fp64, each product and the sum rounded
//     separately (no FMA contraction), truncated toward zero;
//   * the resize is OpenCV's fixed-point INTER_LINEAR for 8-bit images (imgproc/resize.cpp: 11-bit weights from
//     float(double((d + 0.5) * scale - 0.5)), round-half-even; horizontal pass in int, vertical pass
//     (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; an exact 2 x 2 decimation is INTER_AREA: (sum + 2) >> 2).
// The three steps commute into one gather: an output pixel reads its 2 x 2 source pixels (channel 2 - c: BGR -> RGB), each
// blended on the fly when it lies under the cursor.  HBM-bound: H*W*3 bytes in (the touched rows; 691 KB at 640 x 360),
// 48 KB out per frame.  One workgroup = one output row of one frame.
#include "vpt_common.h"
#include "vpt_kernels.h"

// hipcc defaults to -ffp-contract=fast (and the __dmul_rn / __dadd_rn wrappers of the HIP headers are plain operators compiled
// under that default, so they fuse as well): without this pragma a * b + c becomes one fused multiply-add and the blend / the
// coefficient arithmetic round differently from numpy / OpenCV.  All floating-point arithmetic of this file is written with
// plain operators below the pragma.
#pragma clang fp contract(off)

__device__ __forceinline__ int clip_pixel(const VptClipArgs& a, const uint8_t* __restrict__ frame, int gui, int cx, int cy, int cw, int ch,
                                          int y, int x, int c) {
  int v = frame[((size_t)y * a.W + x) * 3 + c];
  const int ry = y - cy, rx = x - cx;
  if (gui && ry >= 0 && ry < ch && rx >= 0 && rx < cw) {
    const double al = a.cursor_alpha[ry * a.CW + rx];
    const double p0 = (double)v * (1.0 - al);                                  // plain operators: contraction is off for this file
    const double p1 = (double)a.cursor_img[(ry * a.CW + rx) * 3 + c] * al;
    v = (int)(p0 + p1);             // astype(np.uint8) of a value in [0, 255]: truncation
  }
  return v;
}

// source index and the two 11-bit weights of destination index d (cv::resize's coefficient loop)
__device__ __forceinline__ void clip_coef(int d, double scale, int src, bool clamp_index, int& s, int& w0, int& w1) {
  const double pos = ((double)d + 0.5) * scale;
  float f = (float)(pos - 0.5);
  s = (int)floorf(f);
  f = f - (float)s;
  if (clamp_index) {           // columns: the index is clamped here and the fraction dropped; rows keep it and clamp at fetch time
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
  }
  w0 = __float2int_rn((1.f - f) * 2048.f);
  w1 = __float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(128) void vpt_clip_kernel(VptClipArgs a) {
  const int dy = blockIdx.x % a.OH, f = blockIdx.x / a.OH;
  const uint8_t* frame = a.src + (size_t)f * a.H * a.W * 3;
  int gui = 0, cx = 0, cy = 0;
  if (a.cursor) { gui = a.cursor[3 * f]; cx = a.cursor[3 * f + 1]; cy = a.cursor[3 * f + 2]; }
  const int ch = max(0, min(a.H - cy, a.CH)), cw = max(0, min(a.W - cx, a.CW));
  if (ch == 0 || cw == 0) gui = 0;
  uint8_t* out = a.dst + ((size_t)f * a.OH + dy) * a.OW * 3;
  if (a.area2x2) {
    for (int dx = threadIdx.x; dx < a.OW; dx += blockDim.x)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int k = 2 - c;
        const int s = clip_pixel(a, frame, gui, cx, cy, cw, ch, 2 * dy, 2 * dx, k) + clip_pixel(a, frame, gui, cx, cy, cw, ch, 2 * dy, 2 * dx + 1, k) +
                      clip_pixel(a, frame, gui, cx, cy, cw, ch, 2 * dy + 1, 2 * dx, k) + clip_pixel(a, frame, gui, cx, cy, cw, ch, 2 * dy + 1, 2 * dx + 1, k);
        out[dx * 3 + c] = (uint8_t)((s + 2) >> 2);
      }
    return;
  }
  int sy, b0, b1;
  clip_coef(dy, a.scale_y, a.H, false, sy, b0, b1);
  const int y0 = min(max(sy, 0), a.H - 1), y1 = min(max(sy + 1, 0), a.H - 1);
  for (int dx = threadIdx.x; dx < a.OW; dx += blockDim.x) {
    int sx, a0, a1;
    clip_coef(dx, a.scale_x, a.W, true, sx, a0, a1);
    const int x1 = min(sx + 1, a.W - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int k = 2 - c;
      const int h0 = clip_pixel(a, frame, gui, cx, cy, cw, ch, y0, sx, k) * a0 + clip_pixel(a, frame, gui, cx, cy, cw, ch, y0, x1, k) * a1;
      const int h1 = clip_pixel(a, frame, gui, cx, cy, cw, ch, y1, sx, k) * a0 + clip_pixel(a, frame, gui, cx, cy, cw, ch, y1, x1, k) * a1;
      out[dx * 3 + c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
  }
}

extern "C" int vpt_clip_launch(const VptClipArgs* a0, hipStream_t stream) {
  VptClipArgs a = *a0;
  if (a.frames <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0) return -1;
  if (a.cursor && (!a.cursor_img || !a.cursor_alpha || a.CH <= 0 || a.CW <= 0)) return -1;
  // cv::resize: scale = 1 / (dsize / ssize) in double; INTER_LINEAR with an exact 2 x 2 decimation is computed as INTER_AREA
  a.scale_x = 1.0 / ((double)a.OW / (double)a.W);
  a.scale_y = 1.0 / ((double)a.OH / (double)a.H);
  const double eps = 2.220446049250313e-16;
  a.area2x2 = (fabs(a.scale_x - 2.0) < eps && fabs(a.scale_y - 2.0) < eps) ? 1 : 0;
  const long grid = (long)a.frames * a.OH;
  if (grid > 0x7fffffffL) return -2;
  hipLaunchKernelGGL(vpt_clip_kernel, dim3((unsigned)grid), dim3(128), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
