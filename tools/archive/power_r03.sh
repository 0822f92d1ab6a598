#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
sample() { for i in 1 2 3; do echo "$1: $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power|Socket Power" | sed 's/clock level//' | tr '\n' ' ' | cut -c1-200)"; sleep 1.5; done; }
# fused bf16 rollout, all alive, in a loop
cat > /tmp/roll_loop.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import trajopt_grpo_amd as tg
dev = torch.device("cuda", 0)
pol = tg.GaussianActorCritic_NeuralNetwork(20, 4, (256,) * 5, cov=0.3, device=dev)
env = tg.QuadPole(max_steps=256); env.spatial_bounds = tuple((-1e9, 1e9) for _ in range(3))
eng = tg.DeviceRollout(env, pol, 256, 256, seed=1, compute_dtype=torch.bfloat16, fused=True)
t0 = time.time()
while time.time() - t0 < 28:
    for _ in range(20): eng.run()
    torch.cuda.synchronize()
PY
(timeout -k 10 60 python3 /tmp/roll_loop.py > /dev/null 2>&1 &)
sleep 14; sample "fused_rollout(all alive)"; wait; sleep 6
# fp32 chain learner: forward+loss+backward loop, then weight-gradient loop (128 x 4, 1 M rows)
cat > /tmp/f32_loop.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import trajopt_grpo_amd as tg
from trajopt_grpo_amd import mlp as M
dev = torch.device("cuda", 0); which = sys.argv[1]
net = tg.NeuralNetwork(5, 1, (128,) * 4, "ReLU").to(dev)
for p in net.parameters(): p.grad = torch.zeros_like(p)
m = M.GemmMLP(net, torch.float32); rows = 1 << 20
xp = m.prepare_input(torch.randn(rows, 5, device=dev)); act = torch.randn(rows, 1, device=dev); lpo = -torch.rand(rows, device=dev) - 1; adv = torch.randn(rows, device=dev)
fl = lambda: m.forward_loss(xp, 0, act=act, logp_old=lpo, adv=adv, var=torch.full((1,), 0.3), epsilon=0.2, surr_coef=-1.0 / rows)
fl(); saved = (m._acts, m._bits, m._dz_head, m._tmask)
t0 = time.time()
while time.time() - t0 < 24:
    for _ in range(50):
        if which == "fb": fl()
        else:
            m._acts, m._bits, m._dz_head, m._tmask = saved; m._backward_fused_f32()
    torch.cuda.synchronize()
PY
for w in fb dw; do (timeout -k 10 60 python3 /tmp/f32_loop.py $w > /dev/null 2>&1 &); sleep 12; sample "f32_$w(128x4, 1M rows)"; wait; sleep 6; done
