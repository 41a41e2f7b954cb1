"""Import stub for cv2: only resize() on already-128x128 frames (identity) is supported."""
INTER_LINEAR = 1


def resize(img, size, interpolation=None):
    w, h = size
    assert img.shape[0] == h and img.shape[1] == w, "cv2 stub: only identity resize supported"
    return img
