"""Clip data path on the CPU: the oracle's cursor compositing and the product's host logic against outputs of
the LIVE reference (tests/golden/make_golden_clip.py), and the guaranteed properties of the cv2.INTER_LINEAR restatement
(cv2 is not in this image: oracle/clip_oracle.py says "parity unpinned" for that one function)."""
import gzip
import json
import os

import numpy as np
import pytest

import vpt_amd  # noqa: F401
from vpt_amd import clip
from oracle import clip_oracle as C

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = dict(np.load(os.path.join(GOLD, "clip_seed0.npz")))
with gzip.open(os.path.join(GOLD, "clip_actions_seed0.json.gz"), "rt") as fh:
    RECS = json.load(fh)


def _plain(action):
    return {k: (v.tolist() if isinstance(v, np.ndarray) else int(v)) for k, v in action.items()}


def test_oracle_composite_matches_live_reference():
    cur = G["cursor_bgra"]
    alpha, image = cur[:, :, 3:] / 255.0, cur[:, :, :3]
    frames = G["frames"].copy()
    for k, (x, y) in enumerate(G["positions"]):
        C.composite_images_with_alpha(frames[k], image, alpha, int(x), int(y))
    assert np.array_equal(frames, G["composited"])
    assert (G["composited"] != G["frames"]).any(axis=(1, 2, 3)).sum() >= 6       # inside / clipped positions did change pixels
    assert np.array_equal(G["composited"][5], G["frames"][5])                    # x = 64 = W: nothing to draw


def test_json_action_to_env_action_matches_live_reference():
    n_null = 0
    for rec in RECS:
        for step, want in zip(rec["steps"], rec["per_step"]):
            action, null = clip.json_action_to_env_action(step)
            assert list(action.keys()) == list(want["action"].keys())            # key order of NOOP_ACTION
            assert _plain(action) == want["action"] and null == want["null"]
            assert action["camera"].dtype.kind == want["camera_dtype_kind"] == "i"   # the reference's integer camera array
            n_null += null
    assert n_null >= 10


@pytest.mark.parametrize("r", range(3))
def test_clip_steps_matches_the_reference_loop(r):
    rec = RECS[r]
    steps_before = json.dumps(rec["steps"])
    out = clip.clip_steps(rec["steps"], rec["frame_height"])
    assert json.dumps(rec["steps"]) == steps_before                              # the caller's recording is not edited
    want = rec["loop"]
    assert out.keep.tolist() == [w["index"] for w in want] and len(out.actions) == len(want)
    assert out.cursor_state.dtype == np.int32 and out.cursor_state.shape == (len(want), 3)
    for k, w in enumerate(want):
        assert _plain(out.actions[k]) == w["action"]
        assert bool(out.cursor_state[k, 0]) == w["gui"]
        if w["gui"]:
            assert out.cursor_state[k, 1:].tolist() == w["cursor"]
    if r == 1:   # the recording that starts with a stuck attack button: attack stays 0 until it is pressed anew
        first_new = next(i for i, s in enumerate(rec["steps"]) if i > 0 and 0 in s["mouse"]["newButtons"])
        assert all(w["action"]["attack"] == 0 for w in want if w["index"] < first_new) and first_new >= 6


def test_negative_cursor_position_is_rejected():
    step = dict(RECS[0]["steps"][0])
    step = json.loads(json.dumps(step))
    step["isGuiOpen"] = True
    step["mouse"]["x"] = -30.0
    step["keyboard"]["keys"] = ["key.keyboard.w"]
    with pytest.raises(ValueError):
        clip.clip_steps([step], 360)


def test_resize_oracle_properties():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (45, 80, 3), dtype=np.uint8)
    assert np.array_equal(C.resize_linear_u8(img, (80, 45)), img)                               # same size: identity
    assert (C.resize_linear_u8(np.full((37, 53, 3), 201, np.uint8), (128, 128)) == 201).all()   # constants survive the fixed point
    # hand-computed 3 x 3 -> 2 x 2 (weights 0.75 / 0.25 = 1536 / 512 of 2048)
    s = np.array([[10, 20, 30], [40, 50, 60], [70, 80, 90]], np.uint8)[:, :, None].repeat(3, 2)
    o = C.resize_linear_u8(s, (2, 2))
    assert o[0, 0, 0] == 20 and o[0, 1, 0] == 35 and o[1, 0, 0] == 65
    # an exact 2 x 2 decimation is INTER_AREA: rounded box average
    big = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    box = (big[0::2, 0::2].astype(int) + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2
    assert np.array_equal(C.resize_linear_u8(big, (48, 32)), box.astype(np.uint8))
    # bilinear is a convex combination: never outside the range of the image; up-scaling keeps the corners
    o = C.resize_linear_u8(img, (128, 128))
    assert o.shape == (128, 128, 3) and o.min() >= img.min() and o.max() <= img.max()
    up = C.resize_linear_u8(img[:8, :8], (32, 32))
    assert np.array_equal(up[0, 0], img[0, 0]) and np.array_equal(up[-1, -1], img[7, 7])
    # transposing the image transposes the result only approximately (two passes, different rounding): |diff| <= 1
    t = C.resize_linear_u8(np.ascontiguousarray(img.transpose(1, 0, 2)), (128, 128)).transpose(1, 0, 2)
    assert np.abs(t.astype(int) - o.astype(int)).max() <= 1


def test_process_frame_composes_the_three_steps():
    cur = G["cursor_bgra"]
    alpha, image = cur[:, :, 3:] / 255.0, cur[:, :, :3]
    f = G["frames"][1]
    out = C.process_frame(f, True, 5, 9, image, alpha, resolution=(64, 48))
    assert np.array_equal(out, G["composited"][1][:, :, ::-1])                  # identity resize: composite + BGR -> RGB
    assert np.array_equal(C.process_frame(f, False, 5, 9, image, alpha, resolution=(64, 48)), f[:, :, ::-1])


@pytest.mark.parametrize("h,w,dw,dh", [(360, 640, 128, 128), (720, 1280, 128, 128), (100, 37, 128, 128), (45, 80, 31, 17)])
def test_resize_oracle_geometry_against_an_independent_bilinear(h, w, dw, dh):
    """cv2 is absent, but torch's bilinear F.interpolate(align_corners=False, antialias=False) samples at the same half-pixel
    centres with the same edge clamping as cv2.INTER_LINEAR and is an INDEPENDENT implementation (float arithmetic): the
    fixed-point restatement may differ from its rounded result by at most 1 (11-bit weights, two truncating shifts: a small
    negative bias, ~0.1) -- a wrong source index, a swapped weight or a half-pixel shift would show up as errors of tens."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # smooth the noise a little so that a geometric error cannot hide in it, keep hard edges too
    img[: h // 2] = (np.add.outer(np.arange(h // 2) * 3, np.arange(w) * 2)[:, :, None] % 256).astype(np.uint8)
    got = C.resize_linear_u8(img, (dw, dh)).astype(np.int64)
    ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)
    ref = ref[0].permute(1, 2, 0).numpy()
    diff = got - np.rint(ref)
    assert np.abs(got - ref).max() <= 1.0 + 1e-9 and np.abs(diff).max() <= 1, (np.abs(got - ref).max(), np.abs(diff).max())
    assert abs(diff.mean()) < 0.25 and (diff != 0).mean() < 0.3
