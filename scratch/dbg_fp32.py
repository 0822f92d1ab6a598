import torch
dev='cuda'
torch.manual_seed(0)
for rows in (777, 25353, 1<<20):
    a = torch.randn(rows,256,device=dev); w = torch.randn(256,256,device=dev)*0.06; b=torch.randn(256,device=dev)
    ref = (a.double() @ w.double())
    for name, f in [('a@w', lambda: a@w), ('a@w.t()', lambda: a@w.t().contiguous().t()), ('linear', lambda: torch.nn.functional.linear(a, w.t().contiguous())),
                    ('addmm_act', lambda: torch._addmm_activation(torch.zeros(256,device=dev), a, w))]:
        o = f()
        r = ref if name!='addmm_act' else torch.relu(ref)
        print(rows, name, 'rel L2 err', float((o.double()-r).norm()/r.norm()))
    dz = torch.randn(rows,256,device=dev)
    refw = dz.double().t() @ a.double()
    print(rows, 'dz.t()@a', float(((dz.t()@a).double()-refw).norm()/refw.norm()))
print(torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision())
import os; print({k:v for k,v in os.environ.items() if 'TF32' in k or 'HIPBLAS' in k or 'ROCBLAS' in k or 'TUNABLE' in k})
