"""Rank-count independence of the PRODUCT path on real kernels ("1/2/4/8-GPU runs must agree up to all-reduce
summation order"; algorithms/ppo.py:138-139, grpo.py:115,140).  Two fresh child processes form a gloo group, share cuda:0,
own half the groups each and run rollout + PPO.learn / GRPO.learn; a third child runs the same on one rank.  (RCCL needs one
GPU per rank; the collectives' call sites are the same for both backends.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dist_product_worker.py")
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    """{world: [per-rank result dicts]} for world 1 and 2: children started with subprocess (never a re-exec of this process)."""
    tmp = tmp_path_factory.mktemp("dist")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    procs, outs = [], {}
    for world in (1, 2):
        port = _free_port()
        outs[world] = [str(tmp / f"w{world}_r{r}.pt") for r in range(world)]
        for r in range(world):
            procs.append(subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), outs[world][r]],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env))
    for p in procs:
        try:
            log, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank did not finish in 420 s")
        assert p.returncode == 0, log.decode("utf-8", "replace")[-3000:]
    return {w: [torch.load(f, weights_only=False) for f in files] for w, files in outs.items()}


@pytest.mark.parametrize("case", ["ppo_bf16_full", "ppo_f32_minibatch_equal", "ppo_f32_minibatch_ragged", "grpo_bf16", "grpo_f32"])
def test_two_ranks_equal_one_rank_on_the_product_path(runs, case):
    """Trajectory shards bit-for-bit, PPO's global moments, optimizer-step counts, both ranks bit-identical to each other, post-step
    weights within 1e-6 (fp32) / the stated bf16 bound of the one-rank run: dist_product_worker.check_case."""
    sys.path.insert(0, HERE)
    from dist_product_worker import check_case
    worst = check_case(runs[1][0][case], [r[case] for r in runs[2]], case)
    print(case, worst)
