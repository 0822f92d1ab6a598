"""MI355X-native drop-in for the rollout -> returns -> GRPO/PPO update path of
Dyllon-Preston/trajopt-grpo: same `Env / Policy / RolloutManager / Rollout_Buffer / GRPO / PPO /
Pipeline` Python surface, backed by hand-written HIP kernels (gfx950) behind the C ABI in
include/trajopt_grpo_hip.h -- environment dynamics, returns / advantages, the MLP's forward, loss head,
backward and weight-gradient passes on the matrix cores (bf16 chains at 128 / 256 wide, fp32 chains at
64 / 128 / 256 wide: every net shape the reference's factories build), the optimizer step -- and one RCCL
gradient all-reduce per optimizer step.  PyTorch supplies device memory, streams and torch.distributed;
nets outside the kernels' shapes fall back to hipBLASLt GEMMs through torch and say so on the
`trajopt_grpo_amd` logger.  Import as `trajopt_grpo_amd` (the directory name carries a hyphen).
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
