import sys, time, os, shutil, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trajopt_grpo_amd as tg
N_ = tg._native
lib = os.environ.get('ABL')
if lib:
    N_.LIB_PATH = os.path.join(os.path.dirname(N_.LIB_PATH), lib); N_._lib = None
dev=torch.device('cuda',0)
torch.manual_seed(0)
pol = tg.GaussianActorCritic_NeuralNetwork(20,4,(256,)*5,cov=0.3,device=dev)
T=256
env = tg.QuadPole(max_steps=T); env.spatial_bounds = tuple((-1e9,1e9) for _ in range(3))
eng = tg.DeviceRollout(env, pol, 256, 256, seed=1, compute_dtype=torch.bfloat16, fused=True)
eng.run(); torch.cuda.synchronize()
best=1e9
for _ in range(3):
    t=time.perf_counter(); tr=eng.run(); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t)
print(lib or 'full', f"{best*1e3:.2f} ms  {best*1e6/T:.1f} us/step  alive-steps {tr.env_steps()}")
