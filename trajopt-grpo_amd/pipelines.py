"""Pipeline orchestrator and the four factories, same surface as the reference.

Mirrors pipelines/pipeline.py:7-213 and pipelines/{cartpole_pipeline_grpo,cartpole_pipeline_ppo,
quadpole_pipeline_ppo,quadpole2d_pipeline_ppo}.py (same default hyper-parameters).  Components are
duck-typed exactly as in the reference, so any of them can be swapped.  The matplotlib `Dashboard`
and the GIF/Markdown `Publisher` are out of scope: the factories default them to None, which the
orchestrator supports the same way the reference does (`pipeline.py:90,135,167-172,209`).
Checkpoint layout: ./archive/<env>/<test>/<ckpt>/{policy.pt, optimizer.pt|pth, reward.csv, metadata.json}.
"""
from __future__ import annotations

import datetime
import json
import os
from typing import Any, Callable, Dict, Optional

import torch

from .algorithms import GRPO, PPO
from .buffers import Rollout_Buffer
from .distributed import rank_world
from .environments import CartPole, QuadPole, QuadPole2D
from .policies import GaussianActor_NeuralNetwork, GaussianActorCritic_NeuralNetwork
from .rollout import RolloutManager, RolloutWorker


class Pipeline:
    def __init__(self, test_name: str, checkpoint_name: str, env_fn: Callable[[], Any], policy: Any, algorithm: Any,
                 rollout_manager: Any, buffer: Any, visualizer: Optional[Any], publisher: Any,
                 logger: Optional[Any] = None, load_path: Optional[str] = None, save_freq: int = 10,
                 render_freq: int = 40) -> None:
        self.test_name, self.checkpoint_name = test_name, checkpoint_name
        self.env_fn = env_fn
        self.env = env_fn()
        self.env_name = self.env.env_name
        self.policy, self.algorithm = policy, algorithm
        self.rollout_manager, self.buffer = rollout_manager, buffer
        self.visualizer, self.publisher, self.logger = visualizer, publisher, logger
        self.load_path, self.save_freq, self.render_freq = load_path, save_freq, render_freq
        self.today = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
        if load_path is not None:
            self.load()
        self.initialize()

    def initialize(self) -> None:
        self.archive_path = os.path.join(".", "archive", self.env_name, self.test_name, self.checkpoint_name)
        self.publish_path = os.path.join(".", "reports", self.env_name, self.test_name, self.checkpoint_name)
        os.makedirs(self.archive_path, exist_ok=True)
        if self.load_path is not None:
            self.load_metadata(os.path.join(self.load_path, "metadata.json"))
        metadata = self.get_metadata()
        if self.visualizer is not None:
            # the reference's Dashboard reads only the first max_episodes_per_render episodes of each group
            # (visualize/dashboard.py:206-217): copy just those from the device when it renders
            # (scoped to the render call, `_render`: other readers of buffer.group_* keep the whole trajectory)
            self.visualizer.initialize(metadata)

    def load(self) -> None:
        if self.load_path is not None:
            self.algorithm.load(self.load_path)
            self.policy.load(self.load_path)
            self.buffer.load(self.load_path)
            sync = getattr(self.algorithm, "sync_old_policy", None)
            if sync is not None:
                sync()                      # the algorithm copied the policy before the checkpoint was loaded into it

    def save(self, path: str) -> None:
        """One process per GPU: the weights are identical on every rank after the gradient all-reduce, so rank 0
        alone writes the checkpoint (torch.save is not atomic); the others wait for it."""
        rank, world = rank_world()
        if rank == 0:
            self.algorithm.save(path)
            self.policy.save(path)
            self.buffer.save(path)
            with open(os.path.join(path, "metadata.json"), "w") as f:
                json.dump(self.get_metadata(), f, indent=4)
        if world > 1:
            torch.distributed.barrier()

    def get_metadata(self) -> Dict[str, Any]:
        return {
            "test_name": self.test_name,
            "checkpoint_name": self.checkpoint_name,
            "creation_date": self.today,
            "env_name": self.env_name,
            "policy": self.policy.metadata(),
            "algorithm": self.algorithm.metadata(),
            "buffer": self.buffer.metadata(),
            "visualizer": self.visualizer.metadata() if self.visualizer is not None else {},
            "publisher": self.publisher.metadata() if self.publisher is not None else {},
            "logger": self.logger.metadata() if self.logger is not None else {},
        }

    def load_metadata(self, path: str) -> Dict[str, Any]:
        with open(path, "r") as f:
            return json.load(f)

    def train(self, epochs: int) -> None:
        for epoch in range(epochs):
            self.buffer.sample()
            self.algorithm.learn(self.buffer)
            if hasattr(self.visualizer, "plot"):
                self.visualizer.plot()
            if self.visualizer is not None and epoch % self.render_freq == 0:
                self._render()
            if epoch % self.save_freq == 0:
                self.save(self.archive_path)

    def _render(self) -> None:
        """visualizer.render() with the buffer's lazy CPU view cut to what it draws: the reference's Dashboard reads only the
        first `max_episodes_per_render` episodes of each group (visualize/dashboard.py:206-217) -- a few MB from the device
        instead of the whole trajectory (1.7 GB at C3) -- for the duration of the call only."""
        k = getattr(self.visualizer, "max_episodes_per_render", None)
        if k is not None and hasattr(self.buffer, "limited_view"):
            with self.buffer.limited_view(max_episodes=int(k)):
                self.visualizer.render()
        else:
            self.visualizer.render()

    def test(self) -> None:
        self.buffer.sample()

    def publish(self) -> None:
        os.makedirs(self.publish_path, exist_ok=True)
        self.buffer.sample()
        if self.publisher is not None:
            self.publisher.publish(self.publish_path)
            self.publisher.report(self.publish_path, self.get_metadata())
        self.save(self.publish_path)

    def save_trajectory(self) -> None:
        self.buffer.sample()
        self.buffer.save_trajectory(self.archive_path)

    def shutdown(self) -> None:
        self.rollout_manager.shutdown()
        if self.visualizer is not None:
            self.visualizer.close()
        if self.logger is not None:
            self.logger.close()
        print("\n\nPipeline shutdown complete.")


def _assemble(test_name, checkpoint_name, env_fn, policy, algorithm, rollout_manager, buffer, visualizer, publisher,
              logger, load_path):
    buffer = buffer or Rollout_Buffer(rollout_manager=rollout_manager)
    return Pipeline(test_name=test_name, checkpoint_name=checkpoint_name, env_fn=env_fn, policy=policy,
                    algorithm=algorithm, rollout_manager=rollout_manager, buffer=buffer, visualizer=visualizer,
                    publisher=publisher, logger=logger, load_path=load_path)


def create_cartpole_pipeline_grpo(test_name, checkpoint_name, env_fn=None, policy=None, algorithm=None,
                                  rollout_manager=None, buffer=None, visualizer=None, publisher=None, logger=None,
                                  load_path=None) -> Pipeline:
    """pipelines/cartpole_pipeline_grpo.py:21-105 defaults."""
    env_fn = env_fn or (lambda: CartPole())
    policy = policy or GaussianActor_NeuralNetwork(input_dim=5, output_dim=1, hidden_dims=(128, 128, 128, 128), cov=0.5)
    algorithm = algorithm or GRPO(epsilon=0.15, beta=0.5, gamma=0.5, policy=policy,
                                  optimizer=torch.optim.Adam(policy.parameters(), lr=3e-4), ref_model=None,
                                  updates_per_iter=1)
    rollout_manager = rollout_manager or RolloutManager(env_fn=env_fn, worker_class=RolloutWorker, policy=policy,
                                                        num_workers=10, num_episodes_per_worker=10, restart=False)
    return _assemble(test_name, checkpoint_name, env_fn, policy, algorithm, rollout_manager, buffer, visualizer,
                     publisher, logger, load_path)


def _ppo_pipeline(env_fn, policy, lr, updates, gamma, num_workers, episodes, test_name, checkpoint_name, algorithm,
                  rollout_manager, buffer, visualizer, publisher, logger, load_path):
    algorithm = algorithm or PPO(epsilon=0.2, c1=0.5, kl_coeff=0.5, policy=policy,
                                 optimizer=torch.optim.Adam(policy.parameters(), lr=lr), ref_model=None,
                                 updates_per_iter=updates, gamma=gamma, lam=0.95, entropy=0.01, batch_size=None)
    rollout_manager = rollout_manager or RolloutManager(env_fn=env_fn, worker_class=RolloutWorker, policy=policy,
                                                        num_workers=num_workers, num_episodes_per_worker=episodes)
    return _assemble(test_name, checkpoint_name, env_fn, policy, algorithm, rollout_manager, buffer, visualizer,
                     publisher, logger, load_path)


def create_cartpole_pipeline_ppo(test_name, checkpoint_name, env_fn=None, policy=None, algorithm=None,
                                 rollout_manager=None, buffer=None, visualizer=None, publisher=None, logger=None,
                                 load_path=None) -> Pipeline:
    """pipelines/cartpole_pipeline_ppo.py:21-108 defaults."""
    env_fn = env_fn or (lambda: CartPole())
    policy = policy or GaussianActorCritic_NeuralNetwork(input_dim=5, output_dim=1, hidden_dims=(128, 128, 128), cov=0.5)
    return _ppo_pipeline(env_fn, policy, 2e-4, 24, 0.99, 10, 8, test_name, checkpoint_name, algorithm, rollout_manager,
                         buffer, visualizer, publisher, logger, load_path)


def create_quadpole2d_pipeline_ppo(test_name, checkpoint_name, env_fn=None, policy=None, algorithm=None,
                                   rollout_manager=None, buffer=None, visualizer=None, publisher=None, logger=None,
                                   load_path=None) -> Pipeline:
    """pipelines/quadpole2d_pipeline_ppo.py defaults."""
    env_fn = env_fn or (lambda: QuadPole2D())
    policy = policy or GaussianActorCritic_NeuralNetwork(input_dim=10, output_dim=2, hidden_dims=(128, 128, 128), cov=0.5)
    return _ppo_pipeline(env_fn, policy, 2e-4, 24, 0.99, 10, 8, test_name, checkpoint_name, algorithm, rollout_manager,
                         buffer, visualizer, publisher, logger, load_path)


def create_quadpole_pipeline_ppo(test_name, checkpoint_name, env_fn=None, policy=None, algorithm=None,
                                 rollout_manager=None, buffer=None, visualizer=None, publisher=None, logger=None,
                                 load_path=None) -> Pipeline:
    """pipelines/quadpole_pipeline_ppo.py:21-109 defaults."""
    env_fn = env_fn or (lambda: QuadPole())
    policy = policy or GaussianActorCritic_NeuralNetwork(input_dim=20, output_dim=4,
                                                         hidden_dims=(256, 256, 256, 256, 256), cov=0.3)
    return _ppo_pipeline(env_fn, policy, 3e-4, 32, 0.999, 10, 5, test_name, checkpoint_name, algorithm, rollout_manager,
                         buffer, visualizer, publisher, logger, load_path)
