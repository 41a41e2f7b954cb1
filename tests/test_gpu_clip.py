"""Clip data path on the device: vpt_clip_frames through vpt_amd.clip, BIT-EXACT against the oracle
(oracle/clip_oracle.py) and against the live reference's composite_images_with_alpha golden vectors."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd import clip, ops  # noqa: E402
from oracle import clip_oracle as C  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = dict(np.load(os.path.join(GOLD, "clip_seed0.npz")))
DEV = "cuda"


def _oracle_batch(frames, state, cursor_bgra, out_wh):
    alpha, image = cursor_bgra[:16, :16, 3:] / 255.0, cursor_bgra[:16, :16, :3]
    return np.stack([C.process_frame(frames[k], bool(state[k, 0]), int(state[k, 1]), int(state[k, 2]), image, alpha, resolution=out_wh)
                     for k in range(len(frames))])


def test_composite_matches_live_reference_golden():
    """Identity resize isolates the compositing: kernel output (RGB) == the reference's composited frames (BGR) channel-swapped."""
    proc = clip.ClipFrameProcessor(G["cursor_bgra"], device=DEV, resolution=(64, 48))
    state = np.concatenate([np.ones((len(G["positions"]), 1), np.int32), G["positions"].astype(np.int32)], axis=1)
    out = proc(torch.from_numpy(G["frames"]).to(DEV), state)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), G["composited"][:, :, :, ::-1])
    off = state.copy(); off[:, 0] = 0                                            # GUI closed: no cursor
    assert np.array_equal(proc(torch.from_numpy(G["frames"]).to(DEV), off).cpu().numpy(), G["frames"][:, :, :, ::-1])


@pytest.mark.parametrize("h,w,out_wh", [(360, 640, (128, 128)), (720, 1280, (128, 128)), (256, 256, (128, 128)), (64, 64, (128, 128)),
                                        (128, 128, (128, 128)), (100, 37, (128, 128)), (45, 80, (31, 17)), (2, 2, (5, 3))])
def test_frames_bit_exact_vs_oracle(h, w, out_wh):
    rng = np.random.default_rng(h * 1000 + w)
    n = 6 if h * w > 300000 else 12
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames[0] = 255; frames[1] = 0                                               # saturated frames
    cursor = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8)
    state = np.zeros((n, 3), np.int32)
    state[:, 0] = rng.random(n) < 0.7
    state[:, 1] = rng.integers(0, w + 8, n)                                      # some cursors clipped by / beyond the right and bottom edges
    state[:, 2] = rng.integers(0, h + 8, n)
    state[2] = (1, 0, 0); state[3] = (1, max(w - 3, 0), max(h - 2, 0))
    proc = clip.ClipFrameProcessor(cursor, device=DEV, resolution=out_wh)
    out = proc(torch.from_numpy(frames).to(DEV), state)
    torch.cuda.synchronize()
    assert out.dtype == torch.uint8 and tuple(out.shape) == (n, out_wh[1], out_wh[0], 3)
    want = _oracle_batch(frames, state, cursor, out_wh)
    got = out.cpu().numpy()
    assert np.array_equal(got, want), f"{(got != want).sum()} of {got.size} bytes differ, max |diff| {np.abs(got.astype(int) - want).max()}"
    # without cursor states the compositing is skipped altogether
    plain = proc(torch.from_numpy(frames).to(DEV), None).cpu().numpy()
    assert np.array_equal(plain, _oracle_batch(frames, np.zeros((n, 3), np.int32), cursor, out_wh))


def test_recorded_clip_end_to_end():
    """jsonl steps + decoded frames -> kept RGB frames + env actions, as data_loader.py:77-123 emits them."""
    with gzip.open(os.path.join(GOLD, "clip_actions_seed0.json.gz"), "rt") as fh:
        rec = json.load(fh)[0]
    rng = np.random.default_rng(5)
    decoded = rng.integers(0, 256, (len(rec["steps"]), 90, 160, 3), dtype=np.uint8)     # a 160 x 90 "video", one frame per step
    steps = clip.clip_steps(rec["steps"], frame_height=90)
    proc = clip.ClipFrameProcessor(G["cursor_bgra"], device=DEV)
    kept = torch.from_numpy(decoded).to(DEV)[torch.from_numpy(steps.keep).to(DEV)]
    out = proc(kept, steps.cursor_state).cpu().numpy()
    assert out.shape == (len(steps.keep), 128, 128, 3) and len(steps.actions) == len(steps.keep) < len(rec["steps"])
    want = _oracle_batch(decoded[steps.keep], steps.cursor_state, G["cursor_bgra"], (128, 128))
    assert np.array_equal(out, want)
    assert steps.cursor_state[:, 0].sum() > 10                                   # GUI-open steps are in the sample


def test_bad_arguments_fail_loudly():
    proc = clip.ClipFrameProcessor(G["cursor_bgra"], device=DEV)
    frames = torch.zeros(2, 36, 64, 3, dtype=torch.uint8, device=DEV)
    with pytest.raises(ValueError):
        proc(frames, np.array([[1, -1, 0], [0, 0, 0]], np.int32))                # negative cursor position
    with pytest.raises(ValueError):
        proc(frames[..., :2].contiguous(), None)                                 # not 3 channels
    with pytest.raises((TypeError, ValueError)):
        ops.clip_frames(frames.float())                                          # not uint8
    assert tuple(proc(frames[:0], None).shape) == (0, 128, 128, 3)               # empty batch


def test_loader_with_the_device_processor(tmp_path):
    """clip_loader.DataLoader end to end on the GPU: in-memory decoder, real ClipFrameProcessor (cursor sprite from a PNG file)."""
    from PIL import Image
    from vpt_amd import clip_loader
    with gzip.open(os.path.join(GOLD, "clip_actions_seed0.json.gz"), "rt") as fh:
        recs = json.load(fh)
    cur = G["cursor_bgra"]
    Image.fromarray(np.ascontiguousarray(cur[:, :, [2, 1, 0, 3]]), "RGBA").save(tmp_path / "cursor.png")      # RGBA on disk, BGRA in memory
    assert np.array_equal(clip_loader.load_cursor_bgra(str(tmp_path / "cursor.png")), cur)
    rng = np.random.default_rng(9)
    videos = {}
    for k, name in enumerate(("r0", "r1", "r2")):
        steps = recs[k]["steps"][:30]
        (tmp_path / f"{name}.jsonl").write_text("\n".join(json.dumps(s) for s in steps))
        (tmp_path / f"{name}.mp4").write_bytes(b"")
        videos[str(tmp_path / f"{name}.mp4")] = rng.integers(0, 256, (30, 90, 160, 3), dtype=np.uint8)
    import random
    random.seed(1)
    dl = clip_loader.DataLoader(str(tmp_path), n_workers=3, batch_size=3, device=DEV, decoder=lambda p: iter(videos[p]),
                                cursor_file=str(tmp_path / "cursor.png"), chunk_frames=11)
    n, seen = 0, {}
    for frames, actions, ids in dl:
        for frame, action, tid in zip(frames, actions, ids):
            assert frame.is_cuda and frame.dtype == torch.uint8 and tuple(frame.shape) == (128, 128, 3)
            video_path, json_path = dl.demonstration_tuples[tid]
            steps = clip.clip_steps([json.loads(l) for l in open(json_path)], 90)
            k = seen.get(tid, 0)                     # this is the k-th kept step of its recording
            seen[tid] = k + 1
            want = _oracle_batch(videos[video_path][steps.keep[k]:steps.keep[k] + 1], steps.cursor_state[k:k + 1], cur, (128, 128))[0]
            assert np.array_equal(frame.cpu().numpy(), want)
            assert int(action["attack"]) == int(steps.actions[k]["attack"])
            n += 1
    assert n >= 9
