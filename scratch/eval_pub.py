import sys, os, json, torch
sys.path.insert(0,'.')
import trajopt_grpo_amd as tg
dev=torch.device('cuda',0)
for name, env_name, critic, dims in [("cartpole_nn_ppo","CartPole",True,(5,1,(128,128,128))),("quadpole2d_nn_ppo","QuadPole2D",True,(10,2,(128,128,128))),("cartpole_nn_grpo","CartPole",False,(5,1,(128,128,128,128)))]:
    path=os.path.join('tests/golden/published',name)
    cls = tg.GaussianActorCritic_NeuralNetwork if critic else tg.GaussianActor_NeuralNetwork
    pol = cls(dims[0],dims[1],dims[2],cov=0.5,device=dev); pol.load(path)
    for dtype in (torch.float32, torch.float64):
        mgr = tg.RolloutManager(lambda: tg.environments.ENV_CLASSES[env_name](), pol, num_workers=64, num_episodes_per_worker=256, seed=0, dtype=dtype)
        tr = mgr.rollout_device()
        ret = tr.rew.sum(0)
        print(name, dtype, 'avg return %.2f std %.1f mean len %.1f' % (float(ret.mean()), float(ret.std()), float(tr.len.float().mean())))
