"""The UNMODIFIED reference (oracle/_ref/vpt_reference.zip, packaged from /root/reference by oracle/make_ref.py) as an
importable checker  --  TEST INFRASTRUCTURE.

reference() puts the archive (zipimport) and the four import stubs of oracle/ref_stubs on sys.path and returns the reference's own
modules: `agent` (MineRLAgent, agent.py:106-206), `inverse_dynamics_model` (IDMAgent, :21-95), `lib.policy`, ...  The drop-in tests
bind those wrappers to the HIP policy with the single statement INTEGRATION.md documents -- the name the wrapper module imported
from lib.policy is re-pointed to vpt_amd.lib.policy's class -- and run them otherwise untouched.

FakeEnv is what `validate_env` (agent.py:84-97) inspects: `.task.<ENV_KWARGS>` and `.action_space.spaces`.
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference():
    """-> namespace of the unmodified reference's modules, or None when the archive is absent (no /root/reference at build time)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import make_ref
    zip_path = make_ref.reference_path() or make_ref.make(verbose=False)
    if zip_path is None:
        return None
    for p in (os.path.join(ROOT, "oracle", "ref_stubs"), zip_path):
        if p not in sys.path:
            sys.path.insert(0, p)
    import lib.torch_util as tu                      # the reference's own modules from here on
    tu.set_default_torch_device("cpu")               # default_device_type() says "cuda" on a ROCm build even without a GPU
    import agent
    import inverse_dynamics_model
    import lib.policy
    import lib.action_mapping
    return types.SimpleNamespace(agent=agent, idm=inverse_dynamics_model, policy=lib.policy, action_mapping=lib.action_mapping, torch_util=tu)


class FakeEnv:
    """The attributes MineRLAgent.__init__ -> validate_env reads (agent.py:84-97)."""

    def __init__(self, agent_module):
        self.task = types.SimpleNamespace(**agent_module.ENV_KWARGS)
        self.action_space = types.SimpleNamespace(spaces=dict(agent_module.TARGET_ACTION_SPACE))


def model_file_dict(policy_kwargs, pi_head_kwargs):
    """The part of a `.model` pickle run_agent.py:11-14 reads."""
    return {"model": {"args": {"net": {"args": dict(policy_kwargs)}, "pi_head_opts": dict(pi_head_kwargs)}}}
