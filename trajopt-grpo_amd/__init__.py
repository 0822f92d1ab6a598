"""MI355X-native drop-in for the rollout -> returns -> GRPO/PPO update path of
Dyllon-Preston/trajopt-grpo: same `Env / Policy / RolloutManager / Rollout_Buffer / GRPO / PPO /
Pipeline` Python surface, backed by hand-written HIP kernels (gfx950) behind the C ABI in
include/trajopt_grpo_hip.h, PyTorch-ROCm GEMMs for the MLP, and one RCCL gradient all-reduce per
optimizer step.  Import as `trajopt_grpo_amd` (the directory name carries a hyphen).
"""
from . import _native
from .environments import Box, Env, CartPole, Pendulum, QuadPole, QuadPole2D, QuadPoleSwarm, Quadrotor, QuadrotorSwarm
from .policies import (NeuralNetwork, ActorCritic, GaussianActor_NeuralNetwork,
                       GaussianActorCritic_NeuralNetwork)
from .rollout import DeviceRollout, DeviceTrajectory, RolloutManager, RolloutWorker
from .buffers import Buffer, Rollout_Buffer
from .algorithms import Algorithm, GRPO, PPO
from .pipelines import (Pipeline, create_cartpole_pipeline_grpo, create_cartpole_pipeline_ppo,
                        create_quadpole_pipeline_ppo, create_quadpole2d_pipeline_ppo)
from . import distributed, hip_ops

__all__ = [
    "Box", "Env", "CartPole", "Pendulum", "QuadPole", "QuadPole2D", "QuadPoleSwarm", "Quadrotor", "QuadrotorSwarm",
    "NeuralNetwork", "ActorCritic", "GaussianActor_NeuralNetwork", "GaussianActorCritic_NeuralNetwork",
    "DeviceRollout", "DeviceTrajectory", "RolloutManager", "RolloutWorker", "Buffer", "Rollout_Buffer",
    "Algorithm", "GRPO", "PPO", "Pipeline", "create_cartpole_pipeline_grpo", "create_cartpole_pipeline_ppo",
    "create_quadpole_pipeline_ppo", "create_quadpole2d_pipeline_ppo", "distributed", "hip_ops",
]
