"""Drop-in boundary checks that need no GPU: state_dict key set / shapes, parameter count,
state structure, and that the HIP policy refuses to run on CPU instead of silently falling back."""
import pytest
import torch

import vpt_amd  # noqa: F401
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space
from oracle import vpt_oracle as O


@pytest.fixture(scope="module")
def policy_1x():
    return MinecraftAgentPolicy(minecraft_action_space(), O.policy_kwargs_for("1x"), dict(temperature=2.0))


def test_state_dict_matches_reference_keys(policy_1x):
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for("1x"), dict(temperature=2.0))
    spec = {k: tuple(s) for k, s, _ in O.state_dict_spec(cfg)}
    sd = policy_1x.state_dict()
    assert set(sd) == set(spec)
    for k, v in sd.items():
        assert tuple(v.shape) == spec[k], k
    assert sum(p.numel() for p in policy_1x.parameters()) == 70_998_654
    missing, unexpected = policy_1x.load_state_dict(O.synthetic_state_dict(cfg, 0), strict=False)
    assert not missing and not unexpected


def test_initial_state_structure(policy_1x):
    st = policy_1x.initial_state(3)
    assert len(st) == 4
    for m, (k, v) in st:
        assert m is None and k.shape == (3, 128, 1024) and v.dtype == torch.float32 and float(k.abs().sum()) == 0


def test_cpu_forward_refuses(policy_1x):
    img = torch.zeros(1, 1, 128, 128, 3, dtype=torch.uint8)
    with pytest.raises(RuntimeError):
        policy_1x({"img": img}, torch.zeros(1, 1, dtype=torch.bool), policy_1x.initial_state(1))


def test_action_head_algebra():
    from vpt_amd.lib.action_head import make_action_head
    head = make_action_head(minecraft_action_space(11, 7), 16)
    lp = {"buttons": torch.log_softmax(torch.randn(2, 3, 1, 11), -1), "camera": torch.log_softmax(torch.randn(2, 3, 1, 7), -1)}
    ac = head.sample(lp, deterministic=True)
    assert ac["buttons"].shape == (2, 3, 1) and ac["buttons"].dtype == torch.int64
    logp = head.logprob(ac, lp)
    ref = lp["buttons"].max(-1).values.sum(-1) + lp["camera"].max(-1).values.sum(-1)
    assert torch.allclose(logp, ref)
    assert torch.allclose(head.kl_divergence(lp, lp), torch.zeros(2, 3, 1), atol=1e-6)


def test_idm_engine_has_every_attribute_the_shared_cnn_code_reads():
    """IDMEngine borrows PolicyEngine's CNN walk (`_cnn_chunk`): every `self.<attribute>` that code reads must exist on an IDMEngine
    (a missing one only shows on the GPU, in the first IDM forward)."""
    import inspect, re
    from vpt_amd import configs, engine
    from vpt_amd.lib.policy import InverseActionPolicy
    pol = InverseActionPolicy(minecraft_action_space(), pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=configs.idm_kwargs_for("tiny"))
    src = inspect.getsource(engine.PolicyEngine._cnn_chunk)
    attrs = set(re.findall(r"self\.([A-Za-z_][A-Za-z_0-9]*)", src))
    missing = [a for a in sorted(attrs) if not hasattr(pol._engine, a)]
    assert not missing, f"IDMEngine lacks {missing}"


def test_act_keep_record_views():
    """ops.unpack_act_keep: the acting step's packed int64 record decodes through views only (no kernels) -- layout of vpt_act_epilogue."""
    import struct
    from vpt_amd import ops
    f2i = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    rows = [(5957, 60, -3.25, 1.5, -0.75), (0, 120, -0.001, -2.0, 8.0)]
    s64 = lambda u: u - (1 << 64) if u >= (1 << 63) else u          # the record is int64: a set sign bit of the upper float wraps
    keep = torch.tensor([[b, c, f2i(lp), s64(f2i(vd) | (f2i(v) << 32))] for b, c, lp, vd, v in rows], dtype=torch.int64)
    ab, ac, lp, vd, v = ops.unpack_act_keep(keep)
    assert ab.tolist() == [5957, 0] and ac.tolist() == [60, 120]
    assert torch.equal(lp, torch.tensor([-3.25, -0.001])) and torch.equal(vd, torch.tensor([1.5, -2.0])) and torch.equal(v, torch.tensor([-0.75, 8.0]))
    assert lp.data_ptr() - keep.data_ptr() == 16 and vd.data_ptr() - keep.data_ptr() == 24      # views of the record, no copies


def test_named_split_k_is_a_function_of_the_layer_only():
    """under the named tilings the K summation order of a linear may depend on (tiling, splitk, N, K) but never on the row count.
    ops.nk_splitk -- what IDMEngine names -- has no M argument at all; its values for the 4x IDM's layers are pinned here."""
    import inspect
    from vpt_amd import ops
    assert list(inspect.signature(ops.nk_splitk).parameters) == ["n", "k"]
    assert ops.nk_splitk(12288, 4096) == 2          # QKV: 96 column tiles
    assert ops.nk_splitk(4096, 4096) == 8           # proj: 32 tiles
    assert ops.nk_splitk(16384, 4096) == 1          # mlp0: 128 tiles fill the chip already
    assert ops.nk_splitk(4096, 16384) == 8          # mlp1
    assert ops.nk_splitk(40, 4096) == 8 and ops.nk_splitk(4096, 256) == 1
    src = inspect.getsource(ops.linear)
    assert "tl == 0" in src and "nk_splitk" in src  # the M-based automatic split is reachable under tiling="auto" only
