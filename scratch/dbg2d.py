import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import trajopt_grpo_amd as tg
from conftest import load_golden
from test_gpu_parity import native_step
dev = torch.device('cuda', 0)
for name in ['QuadPole2D', 'QuadPole']:
    g = load_golden(f"env_step_{name.lower()}.npz")
    nx, rw, tr, sp, tb = native_step(tg, name, g["state"], g["action"], g["steps"], g["time_balanced"], int(g["max_steps"]), torch.float64, dev)
    d = np.abs(nx - g["next_state"])
    print(name, 'per-column max diff', d.max(0))
    print('reward diff', np.abs(rw - g['reward']).max(), 'trunc eq', np.array_equal(tr, g['truncated']))
    bad = np.argwhere(d > 1e-11)
    print(bad[:10])
    for i, j in bad[:5]:
        print(i, j, nx[i, j], g['next_state'][i, j], g['state'][i], g['action'][i])
