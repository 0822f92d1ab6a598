"""Gaussian actor / actor-critic policies with the reference's surface, resident on the GPU.

Mirrors models/neural_network.py:4-77 and policies/actor_critic.py:73-215, :220-378:
same constructor arguments, `forward / log_prob / value / parameters / state_dict /
load_state_dict / save / load / metadata`, attributes `actor`, `critic`, `cov`, and the
same checkpoint formats (`policy.pt`: bare actor state_dict for the actor-only policy,
`{'actor','critic'}` for the actor-critic).  The MLP GEMMs stay on PyTorch-ROCm
(hipBLASLt -> MFMA); sampling, log-prob and the loss head are HIP kernels (rollout.py,
algorithms.py).  The covariance is a fixed diagonal matrix, never learned
(actor_critic.py:100-103, :247-250).
"""
from __future__ import annotations

import math
import os
from typing import Union

import numpy as np
import torch


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class NeuralNetwork(torch.nn.Module):
    """Sequential(Linear, act, ..., Linear).  models/neural_network.py:4-77
    (parameter names `network.{0,2,...}.{weight,bias}` match the reference checkpoints)."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dims: list, activation: Union[str, list] = "ReLU"):
        super().__init__()
        self.input_dim, self.output_dim, self.hidden_dims = input_dim, output_dim, hidden_dims
        if hidden_dims:
            if isinstance(activation, str):
                activations = [activation] * len(hidden_dims)
            elif isinstance(activation, list):
                assert len(activation) == len(hidden_dims), \
                    "Number of activation functions must equal the number of hidden layers."
                activations = activation
            else:
                raise TypeError("activation must be either a string or a list of strings.")
            dims = [input_dim] + list(hidden_dims)
            layers = []
            for i in range(len(hidden_dims)):
                layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
                layers.append(getattr(torch.nn, activations[i])())
            layers.append(torch.nn.Linear(dims[-1], output_dim))
            self.network = torch.nn.Sequential(*layers)
        else:
            self.network = torch.nn.Sequential(torch.nn.Linear(input_dim, output_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.network(x)


class ActorCritic:
    """policies/actor_critic.py:9-26."""

    def __call__(self, state):
        return self.forward(state)


class _GaussianBase(ActorCritic):
    has_critic = False

    def __init__(self, input_dim, output_dim, hidden_dims, activation="ReLU", cov=0.1, device=None):
        self.input_dim, self.output_dim = input_dim, output_dim
        self.hidden_dims, self.activation = hidden_dims, activation
        self.device = torch.device(device) if device is not None else default_device()
        if isinstance(cov, list):
            self.cov = torch.diag(torch.tensor(cov, dtype=torch.float32))        # actor_critic.py:100-103
        else:
            self.cov = torch.diag(torch.tensor([cov] * output_dim, dtype=torch.float32))
        self.actor = NeuralNetwork(input_dim, output_dim, hidden_dims, activation).to(self.device)
        self.critic = None

    # ---- helpers ----------------------------------------------------------
    @property
    def var(self) -> torch.Tensor:
        """diag(cov) as a CPU float32 vector."""
        return torch.diagonal(self.cov).clone()

    def to(self, device):
        self.device = torch.device(device)
        self.actor.to(self.device)
        if self.critic is not None:
            self.critic.to(self.device)
        return self

    def _prep(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x).float()
        return x.to(self.device, torch.float32)

    def _logp(self, mean, action):
        var = self.var.to(mean.device)
        k = self.output_dim
        quad = (((action - mean) ** 2) / var).sum(-1)
        return -0.5 * quad - 0.5 * k * math.log(2 * math.pi) - 0.5 * torch.log(var).sum()

    def _entropy(self, shape, device):
        var = self.var
        h = 0.5 * self.output_dim * (1.0 + math.log(2 * math.pi)) + 0.5 * float(torch.log(var).sum())
        return torch.full(shape, h, dtype=torch.float32, device=device)

    # ---- reference surface ---------------------------------------------------
    def forward(self, state):
        """actor_critic.py:107-138 / :255-289: sample a ~ N(actor(state), cov).
        Returns (action ndarray float32, log_prob Tensor, value Tensor|None)."""
        state = self._prep(state)
        mean = self.actor(state)
        with torch.no_grad():
            std = torch.sqrt(self.var).to(mean.device)
            action = mean + std * torch.randn(mean.shape, device=mean.device)
        log_prob = self._logp(mean, action)
        value = self.critic(state) if self.critic is not None else None
        return action.detach().cpu().numpy(), log_prob, value

    def log_prob(self, observation, action):
        """actor_critic.py:140-160 / :291-311 -> (log_prob, entropy)."""
        observation, action = self._prep(observation), self._prep(action)
        mean = self.actor(observation)
        return self._logp(mean, action), self._entropy(mean.shape[:-1], mean.device)

    def metadata(self):
        return {
            "input_dim": self.input_dim,
            "output_dim": self.output_dim,
            "hidden_dims": self.hidden_dims,
            "activation": self.activation,
            "cov": self.cov.tolist() if isinstance(self.cov, torch.Tensor) else self.cov,
            "num_parameters": sum(p.numel() for p in self.parameters()),
        }


class GaussianActor_NeuralNetwork(_GaussianBase):
    """policies/actor_critic.py:73-215."""

    def value(self, state):
        return [None] * state.shape[0]                                    # :162-173

    def parameters(self):
        return self.actor.parameters()

    def state_dict(self):
        return self.actor.state_dict()

    def load_state_dict(self, state_dict):
        self.actor.load_state_dict(state_dict)

    def save(self, path):
        torch.save({k: v.cpu() for k, v in self.actor.state_dict().items()}, os.path.join(path, "policy.pt"))

    def load(self, path):
        """The reference has no `load` here, so GRPO resume raises; added."""
        self.actor.load_state_dict(torch.load(os.path.join(path, "policy.pt"), weights_only=True, map_location=self.device))


class GaussianActorCritic_NeuralNetwork(_GaussianBase):
    """policies/actor_critic.py:220-378."""
    has_critic = True

    def __init__(self, input_dim, output_dim, hidden_dims, activation="ReLU", cov=0.1, device=None):
        super().__init__(input_dim, output_dim, hidden_dims, activation, cov, device)
        self.critic = NeuralNetwork(input_dim, 1, hidden_dims, activation).to(self.device)

    def value(self, state):
        return self.critic(self._prep(state)).squeeze()                   # :313-323

    def parameters(self):
        return list(self.actor.parameters()) + list(self.critic.parameters())

    def state_dict(self):
        return {"actor": self.actor.state_dict(), "critic": self.critic.state_dict()}

    def load_state_dict(self, state_dict):
        self.actor.load_state_dict(state_dict["actor"])
        self.critic.load_state_dict(state_dict["critic"])

    def load(self, path):
        sd = torch.load(os.path.join(path, "policy.pt"), weights_only=True, map_location=self.device)
        self.load_state_dict(sd)

    def save(self, save_path):
        sd = {k: {n: v.cpu() for n, v in d.items()} for k, d in self.state_dict().items()}
        torch.save(sd, os.path.join(save_path, "policy.pt"))
