import sys, torch
sys.path.insert(0,'.')
import trajopt_grpo_amd as tg
dev='cuda'
for cd in (torch.float32, torch.bfloat16):
  for dims in [(20,4,(256,256,256)), (20,4,(256,)), (5,1,(128,64))]:
    S,A,hidden=dims
    torch.manual_seed(4)
    net = tg.NeuralNetwork(S,A,hidden,'ReLU').to(dev)
    m = tg.mlp.GemmMLP(net, cd)
    rows = 3*8192+777
    X = torch.randn(rows,S,device=dev); g = torch.randn(rows,A,device=dev)
    for p in net.parameters(): p.grad = torch.zeros_like(p)
    out = m.forward(m.prepare_input(X), keep=True); m.backward(g)
    got = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters(): p.grad=None
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=(cd==torch.bfloat16)):
        ref = net(X)
    ref.float().backward(g)
    print(cd, dims, 'out err', float((out-ref.float()).abs().max()))
    for (n,p),a in zip(net.named_parameters(), got):
        print('   ', n, tuple(p.shape), 'rel L2 err', float((a-p.grad).norm()/p.grad.norm()), 'max', float((a-p.grad).abs().max()), float(p.grad.abs().max()))
