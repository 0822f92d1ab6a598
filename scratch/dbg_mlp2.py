import sys, torch, copy
sys.path.insert(0,'.')
import trajopt_grpo_amd as tg
dev='cuda'
S,A,hidden=20,4,(256,256,256)
torch.manual_seed(4)
net = tg.NeuralNetwork(S,A,hidden,'ReLU').to(dev)
net64 = copy.deepcopy(net).double()
m = tg.mlp.GemmMLP(net, torch.float32)
rows = 3*8192+777
X = torch.randn(rows,S,device=dev); g = torch.randn(rows,A,device=dev)
for p in net.parameters(): p.grad = torch.zeros_like(p)
out = m.forward(m.prepare_input(X), keep=True); m.backward(g)
got = [p.grad.clone() for p in net.parameters()]
for p in net.parameters(): p.grad=None
ref = net(X); ref.backward(g)
r64 = net64(X.double()); r64.backward(g.double())
for (n,p),a,p64 in zip(net.named_parameters(), got, net64.parameters()):
    print(n, 'mine vs f64', float((a.double()-p64.grad).norm()/p64.grad.norm()), ' autograd32 vs f64', float((p.grad.double()-p64.grad).norm()/p64.grad.norm()))
