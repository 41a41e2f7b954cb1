"""CPU oracle of the clip data path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, in numpy / plain Python, what happens to one recorded step between the decoded video frame and the
(128 x 128 x 3 uint8 RGB frame, env action) pair the BC loop consumes:

  * composite_images_with_alpha     data_loader.py:34-46      (cursor drawn over the frame while a GUI is open)
  * BGR -> RGB                      data_loader.py:120        (cv2.cvtColor(..., COLOR_BGR2RGB))
  * resize_image                    agent.py:100-103          (cv2.resize(img, (128, 128), interpolation=cv2.INTER_LINEAR))
(The host logic of the same loader -- json_action_to_env_action, stuck attack, hotbar tracking, null filter -- has no oracle of its
own: the product's vpt_amd.clip functions are compared directly with outputs of the LIVE reference, tests/golden/clip_actions_seed0.json.gz.)

PINNING.  composite_images_with_alpha is checked against the LIVE reference function by tests/golden/make_golden_clip.py ->
tests/golden/clip_seed0.npz (tests/test_clip_cpu.py).
cv2.resize is a third-party dependency that is absent from /root/reference and from this image (requirements.txt names
`opencv-python`, unpinned): `resize_linear_u8` restates the published algorithm of OpenCV 4.x
modules/imgproc/src/resize.cpp for CV_8UC3 + INTER_LINEAR (the generic fixed-point path: resizeGeneric_ with
HResizeLinear<uchar,int,short,INTER_RESIZE_COEF_SCALE=2048> and VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>,
including cv::resize's substitution of INTER_AREA for an exact 2 x 2 decimation).  **parity unpinned** for this one
function: there is no golden vector of cv2.resize in the reference and no cv2 here to produce one; the tests pin the
properties the algorithm guarantees (identity, constants, hand-computed small cases), its geometry against an independent
implementation (torch's float bilinear interpolate with half-pixel centres: every byte within 1) and GPU == oracle bit for bit.
Only tests/, __graft_entry__.smoke() and tools/ may import this file."""
import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def composite_images_with_alpha(image1, image2, alpha, x, y):
    """data_loader.py:34-46, in place on image1 (uint8 [H][W][3]); alpha float64 [h][w][1]."""
    ch = max(0, min(image1.shape[0] - y, image2.shape[0]))
    cw = max(0, min(image1.shape[1] - x, image2.shape[1]))
    if ch == 0 or cw == 0:
        return
    a = alpha[:ch, :cw]
    image1[y:y + ch, x:x + cw, :] = (image1[y:y + ch, x:x + cw, :] * (1 - a) + image2[:ch, :cw, :] * a).astype(np.uint8)


def _coefs(dst, src):
    """Per destination index: source index and the two 11-bit fixed-point weights (cv::resize, INTER_LINEAR, 8U)."""
    inv_scale = float(dst) / float(src)
    scale = 1.0 / inv_scale
    ofs = np.zeros(dst, np.int64)
    w = np.zeros((dst, 2), np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)            # double arithmetic, then one rounding to float
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            f, s = np.float32(0), 0
        if s >= src - 1:
            f, s = np.float32(0), src - 1
        c0, c1 = np.float32(1) - f, f
        # saturate_cast<short>(float) = cvRound: round half to even
        w[d, 0] = int(np.clip(np.rint(np.float32(c0 * np.float32(INTER_RESIZE_COEF_SCALE))), -32768, 32767))
        w[d, 1] = int(np.clip(np.rint(np.float32(c1 * np.float32(INTER_RESIZE_COEF_SCALE))), -32768, 32767))
        ofs[d] = s
    return ofs, w, scale


def _coefs_rows(dst, src):
    """Rows: the weights keep their fraction, the row INDEX is clamped when the two rows are fetched (resizeGeneric_Invoker)."""
    inv_scale = float(dst) / float(src)
    scale = 1.0 / inv_scale
    ofs = np.zeros(dst, np.int64)
    w = np.zeros((dst, 2), np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        c0, c1 = np.float32(1) - f, f
        w[d, 0] = int(np.clip(np.rint(np.float32(c0 * np.float32(INTER_RESIZE_COEF_SCALE))), -32768, 32767))
        w[d, 1] = int(np.clip(np.rint(np.float32(c1 * np.float32(INTER_RESIZE_COEF_SCALE))), -32768, 32767))
        ofs[d] = s
    return ofs, w, scale


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize=(width, height), interpolation=cv2.INTER_LINEAR) for uint8 [H][W][C]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    xofs, xa, sx = _coefs(dw, W)
    yofs, yb, sy = _coefs_rows(dh, H)
    eps = np.finfo(np.float64).eps
    if abs(sx - 2.0) < eps and abs(sy - 2.0) < eps:
        # cv::resize: INTER_LINEAR with an exact 2 x 2 decimation is computed as INTER_AREA (resizeAreaFast_, 2 x 2 box, +2 >> 2)
        s = img.astype(np.int64)
        return ((s[0:2 * dh:2, 0:2 * dw:2] + s[0:2 * dh:2, 1:2 * dw:2] + s[1:2 * dh:2, 0:2 * dw:2] + s[1:2 * dh:2, 1:2 * dw:2] + 2) >> 2).astype(np.uint8)
    src = img.astype(np.int64)
    x1 = np.minimum(xofs + 1, W - 1)
    # horizontal pass on every source row: int = S[sx] * a0 + S[sx + 1] * a1
    hrow = src[:, xofs, :] * xa[None, :, 0, None] + src[:, x1, :] * xa[None, :, 1, None]          # [H][dw][C]
    r0 = np.clip(yofs, 0, H - 1)
    r1 = np.clip(yofs + 1, 0, H - 1)
    s0, s1 = hrow[r0], hrow[r1]                                                                 # [dh][dw][C]
    b0, b1 = yb[:, 0][:, None, None], yb[:, 1][:, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)                                                                 # uchar(): values are within 0..255


def process_frame(frame_bgr, gui_open, cursor_x, cursor_y, cursor_bgr, cursor_alpha, resolution=(128, 128)):
    """data_loader.py:113-122 for one kept frame -> RGB uint8 [128][128][3]."""
    frame = np.array(frame_bgr, dtype=np.uint8, copy=True)
    if gui_open:
        composite_images_with_alpha(frame, cursor_bgr, cursor_alpha, cursor_x, cursor_y)
    frame = frame[:, :, ::-1]
    frame = np.asarray(np.clip(frame, 0, 255), dtype=np.uint8)
    return resize_linear_u8(frame, resolution)
