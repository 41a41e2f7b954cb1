"""CPU oracle of the action codec  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A vectorised numpy restatement of the reference's host-side action arithmetic:
  * CameraQuantizer.discretize / undiscretize                    (lib/actions.py:88-108)
  * ActionMapping.factored_buttons_to_groups                     (lib/action_mapping.py:66-104)
  * CameraHierarchicalMapping.from_factored / to_factored        (lib/action_mapping.py:179-219)
Pinned against the live reference by tests/golden/make_golden_actions.py -> tests/golden/actions_seed0.npz
(random and adversarial inputs: every button combination pattern that exercises the tie rules, inventory, camera
null bins, mu-law bin edges).  Only tests/ may import this file."""
import numpy as np

BUTTONS_ALL = (["attack", "back", "forward", "jump", "left", "right", "sneak", "sprint", "use", "drop", "inventory"]
               + [f"hotbar.{i}" for i in range(1, 10)])                                   # lib/actions.py:21-33
IDX = {n: i for i, n in enumerate(BUTTONS_ALL)}
JOINT_INVENTORY = 8640   # len(itertools.product(groups)) = 10*3*3*3*2*2*2*2*2; "inventory" is appended (lib/action_mapping.py:140)


def discretize(xy, maxval=10, binsize=2, mu=10.0, mu_law=True):
    """lib/actions.py:88-98 (np.round = round half to even)."""
    xy = np.clip(np.asarray(xy, dtype=np.float64), -maxval, maxval)
    if mu_law:
        xy = xy / maxval
        xy = np.sign(xy) * (np.log(1.0 + mu * np.abs(xy)) / np.log(1.0 + mu)) * maxval
    return np.round((xy + maxval) / binsize).astype(np.int64)


def undiscretize(pq, maxval=10, binsize=2, mu=10.0, mu_law=True):
    """lib/actions.py:100-108."""
    xy = np.asarray(pq) * binsize - maxval
    if mu_law:
        xy = xy / maxval
        xy = np.sign(xy) * (1.0 / mu) * ((1.0 + mu) ** np.abs(xy) - 1.0) * maxval
    return xy


def from_factored(buttons, camera, n_camera_bins=11):
    """lib/action_mapping.py:179-207 -> (joint buttons [B], joint camera [B])."""
    b = np.asarray(buttons) != 0
    cam = np.asarray(camera)
    null = n_camera_bins // 2
    hotbar = np.zeros(len(b), dtype=np.int64)
    for k in range(1, 10):                                   # later button in the group wins (:99-103)
        hotbar = np.where(b[:, IDX[f"hotbar.{k}"]], k, hotbar)
    both = b[:, IDX["forward"]] & b[:, IDX["back"]]          # mutual press = neither (:93-96)
    fb = np.where(both, 0, np.where(b[:, IDX["back"]], 2, np.where(b[:, IDX["forward"]], 1, 0)))
    both = b[:, IDX["left"]] & b[:, IDX["right"]]
    lr = np.where(both, 0, np.where(b[:, IDX["right"]], 2, np.where(b[:, IDX["left"]], 1, 0)))
    ss = np.where(b[:, IDX["sneak"]], 2, np.where(b[:, IDX["sprint"]], 1, 0))
    cam_on = ~np.all(cam == null, axis=1)
    jb = hotbar
    for digit, radix in ((fb, 3), (lr, 3), (ss, 3), (b[:, IDX["use"]], 2), (b[:, IDX["drop"]], 2), (b[:, IDX["attack"]], 2),
                         (b[:, IDX["jump"]], 2), (cam_on, 2)):
        jb = jb * radix + digit.astype(np.int64)
    jc = cam[:, 0].astype(np.int64) * n_camera_bins + cam[:, 1]
    inv = np.asarray(buttons)[:, IDX["inventory"]] == 1        # (:196-197) exclusive with everything incl. the camera (:200-205)
    return np.where(inv, JOINT_INVENTORY, jb), np.where(inv, null * n_camera_bins + null, jc)


def to_factored(joint_buttons, joint_camera, n_camera_bins=11):
    """lib/action_mapping.py:209-219 (tables of :149-177 decoded arithmetically)."""
    jb = np.asarray(joint_buttons).astype(np.int64).copy()
    jc = np.asarray(joint_camera).astype(np.int64)
    n = len(jb)
    null = n_camera_bins // 2
    out = np.zeros((n, len(BUTTONS_ALL)), dtype=np.int64)
    inv = jb == JOINT_INVENTORY
    out[inv, IDX["inventory"]] = 1
    rest = np.where(inv, 0, jb)
    cam_on = rest % 2; rest //= 2
    for name in ("jump", "attack", "drop", "use"):
        out[:, IDX[name]] = np.where(inv, 0, rest % 2); rest //= 2
    for first, second in (("sprint", "sneak"), ("left", "right"), ("forward", "back")):
        d = rest % 3; rest //= 3
        out[:, IDX[first]] = np.where(inv, 0, d == 1)
        out[:, IDX[second]] = np.where(inv, 0, d == 2)
    for k in range(1, 10):
        out[:, IDX[f"hotbar.{k}"]] = np.where(inv, 0, rest == k)
    cam_off = (~inv) & (cam_on == 0)                           # the table leaves the flag False for "inventory" (:163-168)
    cam = np.stack([jc // n_camera_bins, jc % n_camera_bins], axis=1)
    cam[cam_off] = null
    return out, cam
