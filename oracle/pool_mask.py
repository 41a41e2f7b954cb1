"""TEST INFRASTRUCTURE (oracle/): CPU restatement of the arg-max mask format of the round-5 training forward.

vpt_conv3x3_kernel mode 7 + vpt_pool_seam_kernel store, per pooled value of F.max_pool2d(x, 3, stride 2, padding 1)
(/root/reference/lib/impala_cnn.py:117) over a POST-ReLU tensor x >= 0, a 9-bit word: bit (8 - k) is 1 iff window position
k = 3 (dy + 1) + (dx + 1), dy, dx in {-1, 0, 1}, lies outside the image or holds a value different from the window's maximum
max(0, values).  The backward (vpt_conv_bwd_prep_pooled_kernel) routes the pooled gradient to the FIRST position in scan order
whose bit is 0 -- the rule of torch's max_pool2d backward (first maximum wins a tie) -- and passes nothing where the maximum is 0.
Only tests/ import this module."""
import numpy as np


def pool_argmax_masks(x):
    """x: float array [..., H, W] with x >= 0, H and W even -> (pooled [..., H/2, W/2], masks uint16 of the same shape)."""
    x = np.asarray(x)
    h, w = x.shape[-2:]
    ph, pw = h // 2, w // 2
    pad = np.full(x.shape[:-2] + (h + 2, w + 2), -1.0, dtype=x.dtype)      # -1: below every post-ReLU value and never equal to a maximum >= 0
    pad[..., 1:-1, 1:-1] = x
    wins = [pad[..., dy:dy + h:2, dx:dx + w:2][..., :ph, :pw] for dy in range(3) for dx in range(3)]
    pooled = np.maximum(np.max(np.stack(wins, 0), 0), 0.0)
    masks = np.zeros(pooled.shape, dtype=np.uint16)
    for k, v in enumerate(wins):
        masks |= ((v != pooled).astype(np.uint16) << (8 - k))
    return pooled, masks


def decode_first_position(masks):
    """-> int array: the first scan position whose bit is 0 (0..8)."""
    inv = (~masks.astype(np.uint32)) & 0x1ff
    assert (inv != 0).all(), "a window without a position that holds its maximum"
    return 8 - np.floor(np.log2(inv)).astype(np.int64)
