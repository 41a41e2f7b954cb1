"""Import stub for cv2  --  TEST INFRASTRUCTURE (the image has no opencv-python).
resize(): identity on frames that already have the target size; otherwise oracle/clip_oracle.py's restatement of the 8-bit
cv2.INTER_LINEAR path (parity unpinned against real OpenCV, see that file's header) -- enough for agent.py:100-103 to run
on a 640x360 observation in the drop-in tests."""
INTER_LINEAR = 1


def resize(img, size, interpolation=None):
    w, h = size
    if img.shape[0] == h and img.shape[1] == w:
        return img
    assert interpolation in (None, INTER_LINEAR), "cv2 stub: INTER_LINEAR only"
    from oracle.clip_oracle import resize_linear_u8
    return resize_linear_u8(img, (w, h))
