"""vpt_amd.clip_loader.DataLoader (the reference's data_loader.DataLoader protocol) on the CPU: synthetic recordings, an
in-memory "decoder" and the oracle as frame processor -- the sampling rule (round-robin lanes, next recording when a lane runs
out, stop at the first empty lane), the null-action filter, chunked decoding and short videos."""
import gzip
import json
import os
import random

import numpy as np
import pytest
import torch

import vpt_amd  # noqa: F401
from vpt_amd import clip, clip_loader
from oracle import clip_oracle as C

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = dict(np.load(os.path.join(GOLD, "clip_seed0.npz")))
with gzip.open(os.path.join(GOLD, "clip_actions_seed0.json.gz"), "rt") as fh:
    RECS = json.load(fh)
H, W = 36, 64


def _video(name, n):
    """Deterministic frames: pixel value encodes (recording, frame index)."""
    base = sum(map(ord, name)) % 200
    return [np.full((H, W, 3), (base + 7 * i) % 256, np.uint8) + np.arange(3, dtype=np.uint8) for i in range(n)]


def _oracle_processor(frames, cursor_state):
    cur = G["cursor_bgra"]
    alpha, image = cur[:16, :16, 3:] / 255.0, cur[:16, :16, :3]
    out = [C.process_frame(f.numpy(), bool(s[0]), int(s[1]), int(s[2]), image, alpha) for f, s in zip(frames, cursor_state)]
    return torch.from_numpy(np.stack(out)) if out else torch.zeros(0, 128, 128, 3, dtype=torch.uint8)


@pytest.fixture()
def dataset(tmp_path):
    lengths = {"a": 23, "b": 9, "c": 40, "d": 15, "e": 31}
    videos = {}
    for k, (name, n) in enumerate(lengths.items()):
        steps = RECS[k % len(RECS)]["steps"][:n]
        with open(tmp_path / f"{name}.jsonl", "w") as f:
            f.write("\n".join(json.dumps(s) for s in steps))
        (tmp_path / f"{name}.mp4").write_bytes(b"")                 # only the name is used: the decoder below is in-memory
        videos[str(tmp_path / f"{name}.mp4")] = _video(name, n if name != "d" else n - 4)   # "d": the video ends 4 frames early
    return tmp_path, lengths, videos


def _expected_stream(loader_tuples, videos, n_workers):
    """The reference's rule, written out independently: per-recording kept items, lanes served round-robin."""
    per_task = []
    for tid, (video, jsonl) in enumerate(loader_tuples):
        data = json.loads("[" + ",".join(open(jsonl).readlines()) + "]")
        frames = videos[video]
        steps = clip.clip_steps(data, H)
        items = [(tid, int(i), a) for i, a in zip(steps.keep, steps.actions) if i < len(frames)]
        per_task.append(items)
    lanes, nxt, out, turn = [[] for _ in range(n_workers)], 0, [], 0
    while True:
        lane = lanes[turn % n_workers]
        while not lane and nxt < len(per_task):
            lane.extend(per_task[nxt]); nxt += 1
        if not lane:
            return out
        out.append(lane.pop(0)); turn += 1


@pytest.mark.parametrize("n_workers,batch_size,chunk", [(2, 2, 512), (3, 2, 5), (5, 4, 7)])
def test_loader_stream_matches_the_reference_rule(dataset, n_workers, batch_size, chunk):
    root, lengths, videos = dataset
    random.seed(3)
    with pytest.warns(UserWarning) if True else None:
        dl = clip_loader.DataLoader(str(root), n_workers=n_workers, batch_size=batch_size, n_epochs=2, device="cpu",
                                    decoder=lambda p: iter(videos[p]), frame_processor=_oracle_processor, chunk_frames=chunk)
        got = []
        for frames, actions, ids in dl:
            assert len(frames) == len(actions) == len(ids) == batch_size
            got += list(zip(ids, frames, actions))
    want = _expected_stream(dl.demonstration_tuples, videos, n_workers)
    want = want[: (len(want) // batch_size) * batch_size] if len(got) < len(want) else want
    # iteration stops when the lane whose turn it is is empty: everything before that point, in order
    assert len(got) >= batch_size and len(got) % batch_size == 0 and len(got) <= len(want) + batch_size
    assert len(dl.demonstration_tuples) == 2 * len(lengths)
    cur = G["cursor_bgra"]
    alpha, image = cur[:16, :16, 3:] / 255.0, cur[:16, :16, :3]
    for (tid, frame, action), (wtid, widx, waction) in zip(got, want):
        assert tid == wtid
        assert {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in action.items()} == \
               {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in waction.items()}
        video = videos[dl.demonstration_tuples[tid][0]]
        data = json.loads("[" + ",".join(open(dl.demonstration_tuples[tid][1]).readlines()) + "]")
        st = data[widx]
        cx, cy = (int(st["mouse"]["x"] * H / 720), int(st["mouse"]["y"] * H / 720)) if st["isGuiOpen"] else (0, 0)
        ref = C.process_frame(video[widx], bool(st["isGuiOpen"]), cx, cy, image, alpha)
        assert tuple(frame.shape) == (128, 128, 3) and np.array_equal(np.asarray(frame), ref)


def test_loader_argument_checks(dataset):
    root, lengths, videos = dataset
    with pytest.raises(AssertionError):
        clip_loader.DataLoader(str(root), n_workers=2, batch_size=3, device="cpu", decoder=lambda p: iter(()), frame_processor=_oracle_processor)
    with pytest.raises(AssertionError):
        clip_loader.DataLoader(str(root), n_workers=9, batch_size=2, device="cpu", decoder=lambda p: iter(()), frame_processor=_oracle_processor)


def test_to_numpy_returns_host_arrays(dataset):
    root, lengths, videos = dataset
    random.seed(0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dl = clip_loader.DataLoader(str(root), n_workers=2, batch_size=2, device="cpu", to_numpy=True,
                                    decoder=lambda p: iter(videos[p]), frame_processor=_oracle_processor)
        frames, actions, ids = next(dl)
    assert isinstance(frames[0], np.ndarray) and frames[0].dtype == np.uint8 and frames[0].shape == (128, 128, 3)
    assert set(actions[0]) >= {"camera", "attack", "forward", "ESC"}
