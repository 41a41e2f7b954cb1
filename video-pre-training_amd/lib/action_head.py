"""Categorical / dict action heads over the log-probabilities produced by the HIP path.

Same method surface as lib/action_head.py:136-260 (logprob, sample, entropy, kl_divergence); the linear
layer, temperature and log_softmax of CategoricalActionHead.forward run inside the fused heads GEMM +
vpt_log_softmax_forward, so the classes here only hold the parameters and the distribution algebra."""
from typing import Tuple

import torch
from torch import nn


class CategoricalActionHead(nn.Module):
    def __init__(self, input_dim: int, shape: Tuple[int, ...], num_actions: int, temperature: float = 1.0):
        super().__init__()
        self.input_dim = input_dim
        self.num_actions = num_actions
        self.output_shape = tuple(shape) + (num_actions,)
        self.temperature = temperature
        n_out = 1
        for s in self.output_shape:
            n_out *= s
        self.linear_layer = nn.Linear(input_dim, n_out)

    def logprob(self, actions: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
        """lib/action_head.py:176-184: gather the log-pmf at the action index, sum over the shape dims."""
        value = actions.long().unsqueeze(-1)
        value, log_pmf = torch.broadcast_tensors(value, logits)
        value = value[..., :1]
        result = log_pmf.gather(-1, value).squeeze(-1)
        for _ in self.output_shape[:-1]:
            result = result.sum(dim=-1)
        return result

    def entropy(self, logits: torch.Tensor) -> torch.Tensor:
        ent = -(torch.exp(logits) * logits).sum(dim=-1)
        for _ in self.output_shape[:-1]:
            ent = ent.sum(dim=-1)
        return ent

    def sample(self, logits: torch.Tensor, deterministic: bool = False) -> torch.Tensor:
        """argmax, or Gumbel-max with the u == 1.0 guard of lib/action_head.py:195-207."""
        if deterministic:
            return torch.argmax(logits, dim=-1)
        u = torch.rand_like(logits)
        u[u == 1.0] = 0.999
        return torch.argmax(logits - torch.log(-torch.log(u)), dim=-1)

    def kl_divergence(self, logits_q: torch.Tensor, logits_p: torch.Tensor) -> torch.Tensor:
        kl = (torch.exp(logits_q) * (logits_q - logits_p)).sum(-1, keepdim=True)
        for _ in self.output_shape[:-1]:
            kl = kl.sum(dim=-2)
        return kl


class DictActionHead(nn.ModuleDict):
    def logprob(self, actions, logits):
        return sum(sub.logprob(actions[k], logits[k]) for k, sub in self.items())

    def sample(self, logits, deterministic: bool = False):
        return {k: sub.sample(logits[k], deterministic) for k, sub in self.items()}

    def entropy(self, logits):
        return sum(sub.entropy(logits[k]) for k, sub in self.items())

    def kl_divergence(self, logits_q, logits_p):
        return sum(sub.kl_divergence(logits_q[k], logits_p[k]) for k, sub in self.items())


def make_action_head(ac_space, pi_out_size: int, temperature: float = 1.0):
    """lib/action_head.py:263-275 for the spaces VPT instantiates (dict of discrete tensors)."""
    if hasattr(ac_space, "items"):
        return DictActionHead({k: make_action_head(v, pi_out_size, temperature) for k, v in ac_space.items()})
    if hasattr(ac_space, "eltype") and hasattr(ac_space.eltype, "n"):
        return CategoricalActionHead(pi_out_size, ac_space.shape, ac_space.eltype.n, temperature=temperature)
    raise NotImplementedError(f"Action space of type {type(ac_space)} is not supported")
