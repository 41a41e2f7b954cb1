"""Loader of the contractor dataset: the reference's `data_loader.DataLoader` (data_loader.py:131-222) with the
per-frame pixel work batched onto the GPU  (`from vpt_amd.clip_loader import DataLoader`).

Same constructor arguments, same iteration protocol -- `for batch_frames, batch_actions, batch_episode_id in loader` -- and the
same sampling rule: the stream is drawn round-robin from `n_workers` lanes, one kept (non-null) step at a time, every lane walks
its recording in order and takes the next unassigned recording when it runs out, and iteration stops when the lane whose turn it
is has nothing left (data_loader.py:198-217).  What differs:

  * A lane is not a process that pushes one frame at a time through cv2 and numpy (data_loader.py:48-128).  It decodes a chunk of
    a recording, applies the loader's per-step logic to the jsonl once (`clip.clip_steps`: stuck attack button, hotbar tracking,
    json_action_to_env_action, null-action filter, cursor position), moves ONLY the kept frames to the GPU and runs cursor
    compositing + BGR->RGB + the cv2.INTER_LINEAR resize for all of them in one launch (`clip.ClipFrameProcessor`).
  * The reference's lanes race each other for recordings, so its sample order depends on process timing.  Here the schedule is
    the one that race has when every worker is equally fast: lanes are served strictly round-robin, so a lane asks for its next
    recording exactly when its current one is used up -- deterministic for a given file order.
  * Frames are returned as uint8 RGB tensors [128, 128, 3] on the device (`to_numpy=True`: numpy arrays on the host, what the
    reference returns and `agent._env_obs_to_agent` expects).

Decoding stays on the CPU: `decoder(video_path)` must yield BGR uint8 frames [H, W, 3] in order, as cv2.VideoCapture.read() does;
the default uses cv2 and raises ImportError where cv2 is not installed (it is not part of this project's image)."""
import glob
import json
import os
import random
import warnings
from typing import Callable, Iterator, List, Optional

import numpy as np

from . import clip

CURSOR_FILE = os.path.join("cursors", "mouse_cursor_white_16x16.png")     # data_loader.py:18, relative to the reference checkout


def cv2_decoder(video_path: str) -> Iterator[np.ndarray]:
    """Frames of an mp4 in order, BGR uint8 (cv2.VideoCapture, data_loader.py:62,106)."""
    try:
        import cv2
    except ImportError as e:   # pragma: no cover - cv2 is absent from the build image
        raise ImportError("decoding the contractor recordings needs opencv-python (cv2.VideoCapture); pass decoder=... to use another one") from e
    video = cv2.VideoCapture(video_path)
    try:
        while True:
            ret, frame = video.read()
            if not ret:
                return
            yield frame
    finally:
        video.release()


def load_cursor_bgra(path: str) -> np.ndarray:
    """The cursor sprite as cv2.imread(path, IMREAD_UNCHANGED) returns it: uint8 [h, w, 4], BGRA (data_loader.py:53)."""
    try:
        import cv2
        img = cv2.imread(path, cv2.IMREAD_UNCHANGED)
        if img is None:
            raise FileNotFoundError(path)
        return img
    except ImportError:
        from PIL import Image
        rgba = np.asarray(Image.open(path).convert("RGBA"))
        return np.ascontiguousarray(rgba[:, :, [2, 1, 0, 3]])


class _Lane:
    """One of the reference's worker slots: a cursor over the kept steps of its current recording."""

    def __init__(self, loader):
        self.loader = loader
        self.items = iter(())

    def next_item(self):
        while True:
            item = next(self.items, None)
            if item is not None:
                return item
            task = self.loader._next_task()
            if task is None:
                return None            # the reference's worker puts None when the task queue is empty (data_loader.py:127-128)
            self.items = self.loader._recording_items(*task)


class DataLoader:
    """data_loader.DataLoader (data_loader.py:131-222): `dataset_dir` holds <id>.mp4 + <id>.jsonl pairs."""

    def __init__(self, dataset_dir, n_workers=8, batch_size=8, n_epochs=1, max_queue_size=16, device="cuda", to_numpy=False,
                 decoder: Optional[Callable[[str], Iterator[np.ndarray]]] = None, frame_processor=None, cursor_file: Optional[str] = None,
                 chunk_frames: int = 512):
        assert n_workers >= batch_size, "Number of workers must be equal or greater than batch size"
        self.dataset_dir = dataset_dir
        self.n_workers = n_workers
        self.n_epochs = n_epochs
        self.batch_size = batch_size
        self.max_queue_size = max_queue_size          # kept for signature compatibility: lanes are pulled, not pushed, so nothing queues up
        self.to_numpy = to_numpy
        self.chunk_frames = int(chunk_frames)
        unique_ids = glob.glob(os.path.join(dataset_dir, "*.mp4"))
        unique_ids = sorted(set(os.path.basename(x).split(".")[0] for x in unique_ids))   # sorted: the reference's set() order is arbitrary
        self.unique_ids = unique_ids
        demonstration_tuples = [(os.path.abspath(os.path.join(dataset_dir, u + ".mp4")), os.path.abspath(os.path.join(dataset_dir, u + ".jsonl")))
                                for u in unique_ids]
        assert n_workers <= len(demonstration_tuples), f"n_workers should be lower or equal than number of demonstrations {len(demonstration_tuples)}"
        self.demonstration_tuples = []
        for _ in range(n_epochs):                     # every epoch in a fresh random order (random.shuffle, as the reference: seed `random` to fix it)
            random.shuffle(demonstration_tuples)
            self.demonstration_tuples += demonstration_tuples
        self._tasks = [(trajectory_id, *task) for trajectory_id, task in enumerate(self.demonstration_tuples)]
        self._task_pos = 0
        self.n_steps_processed = 0
        self._decoder = decoder or cv2_decoder
        if frame_processor is None:
            path = cursor_file or CURSOR_FILE
            frame_processor = clip.ClipFrameProcessor(load_cursor_bgra(path), device=device)
        self._processor = frame_processor
        self._device = device
        self._lanes = [_Lane(self) for _ in range(n_workers)]

    # ---- recordings -------------------------------------------------------------------------------------------------
    def _next_task(self):
        if self._task_pos >= len(self._tasks):
            return None
        task = self._tasks[self._task_pos]
        self._task_pos += 1
        return task

    def _recording_items(self, trajectory_id, video_path, json_path):
        """(trajectory_id, frame, action) for every kept step of one recording, in order (data_loader.py:62-123), chunk by chunk."""
        import torch
        with open(json_path) as json_file:
            json_data = json.loads("[" + ",".join(json_file.readlines()) + "]")
        frames = iter(self._decoder(video_path))
        steps = None
        pos = 0                                        # index of the next recorded step = of the next decoded frame
        while pos < len(json_data):
            chunk = []
            for frame in frames:
                chunk.append(frame)
                if len(chunk) >= self.chunk_frames or pos + len(chunk) >= len(json_data):
                    break
            if not chunk:
                # the reference prints this for every step past the end of the video and goes on (data_loader.py:124-125)
                warnings.warn(f"Could not read frame from video {video_path}: {len(json_data) - pos} recorded steps have no frame")
                return
            if steps is None:                          # the per-step logic is stateful over the whole recording: one pass
                steps = clip.clip_steps(json_data, frame_height=chunk[0].shape[0])
            lo, hi = np.searchsorted(steps.keep, [pos, pos + len(chunk)])
            if hi > lo:
                local = steps.keep[lo:hi] - pos
                kept = np.stack([chunk[i] for i in local])                       # only the kept frames cross PCIe
                out = self._processor(torch.from_numpy(kept).to(self._device), steps.cursor_state[lo:hi])
                if self.to_numpy:
                    out = out.cpu().numpy()
                for k in range(hi - lo):
                    yield trajectory_id, out[k], steps.actions[lo + k]
            pos += len(chunk)

    # ---- iteration (data_loader.py:195-217) -----------------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        batch_frames: List = []
        batch_actions: List[dict] = []
        batch_episode_id: List[int] = []
        for _ in range(self.batch_size):
            workitem = self._lanes[self.n_steps_processed % self.n_workers].next_item()
            if workitem is None:
                # stop when the first lane runs out of work, as the reference does: the batches stay diverse
                raise StopIteration()
            trajectory_id, frame, action = workitem
            batch_frames.append(frame)
            batch_actions.append(action)
            batch_episode_id.append(trajectory_id)
            self.n_steps_processed += 1
        return batch_frames, batch_actions, batch_episode_id
