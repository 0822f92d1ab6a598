"""Host-side timeline of a C2 step (CartPole GRPO, 4,096 envs x 500, fp32 5-128-128-1, 10 updates): time.perf_counter stamps around the
learner's prologue enqueue, the one host wait of a step (the row count), and the first chain launch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
from trajopt_grpo_amd import algorithms as A, mlp as M
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pol = tg.GaussianActor_NeuralNetwork(5, 1, (128, 128), cov=0.5, device=dev)
mgr = tg.RolloutManager(lambda: tg.CartPole(max_steps=500), pol, num_workers=64, num_episodes_per_worker=64, seed=1)
buf = tg.Rollout_Buffer(mgr)
algo = tg.GRPO(epsilon=0.15, beta=0.5, gamma=0.99, policy=pol, optimizer=torch.optim.Adam(pol.parameters(), lr=3e-4), updates_per_iter=10)
st = {"wake": [], "first": [], "learn0": [], "enq": [], "learn1": []}
pc = time.perf_counter
orig_pf = A._GpuLearner._prepare_finish
orig_pe = A._GpuLearner._prepare_enqueue
orig_fl = M.GemmMLP._forward_loss_f32
orig_learn = A.GRPO._learn
state = {"n": 0}
def pe(self, *a, **k):
    r = orig_pe(self, *a, **k); st["enq"].append(pc()); return r
def pf(self, h):
    t0 = pc(); r = orig_pf(self, h); st["wake"].append((t0, pc())); state["n"] = 0; return r
def fl(self, *a, **k):
    r = orig_fl(self, *a, **k)
    if state["n"] == 0: st["first"].append(pc())
    state["n"] += 1
    return r
def learn(self, b):
    st["learn0"].append(pc()); r = orig_learn(self, b); st["learn1"].append(pc()); return r
A._GpuLearner._prepare_finish = pf; A._GpuLearner._prepare_enqueue = pe; M.GemmMLP._forward_loss_f32 = fl; A.GRPO._learn = learn
for i in range(80):
    buf.sample(); algo.learn(buf)
torch.cuda.synchronize()
import numpy as np
k = 20
w0 = np.array([a for a, b in st["wake"]][k:]); w1 = np.array([b for a, b in st["wake"]][k:])
first = np.array(st["first"][k:]); l0 = np.array(st["learn0"][k:]); l1 = np.array(st["learn1"][k:]); enq = np.array(st["enq"][k:])
print("learn entry -> prologue enqueued   %.1f us" % (1e6 * (enq - l0).mean()))
print("prologue enqueued -> finish called  %.1f us" % (1e6 * (w0 - enq).mean()))
print("inside _prepare_finish (the wait)   %.1f us" % (1e6 * (w1 - w0).mean()))
print("wake -> first chain launch returned %.1f us" % (1e6 * (first - w1).mean()))
print("first launch -> learn() returns     %.1f us" % (1e6 * (l1 - first).mean()))
print("learn() total %.1f us; step total %.1f us" % (1e6 * (l1 - l0).mean(), 1e6 * np.diff(l0).mean()))
