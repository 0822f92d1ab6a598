import torch, time
dev='cuda'
rows, K, N = 1<<20, 256, 256
x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
dy = torch.randn(rows, N, device=dev, dtype=torch.bfloat16)
def timeit(f, n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
print('plain dY^T X us', timeit(lambda: dy.t() @ x))
for B in (64, 128, 256, 512, 1024):
    r = rows//B
    def f():
        p = torch.bmm(dy.view(B, r, N).transpose(1,2), x.view(B, r, K))
        return p.sum(0, dtype=torch.float32)
    print('bmm split', B, timeit(f))
    try:
        def g():
            p = torch.bmm(dy.view(B, r, N).transpose(1,2), x.view(B, r, K), out_dtype=torch.float32)
            return p.sum(0)
        print('bmm split fp32 out', B, timeit(g))
    except Exception as e:
        print('out_dtype failed', repr(e)[:200])
for (kk, nn) in ((20, 256), (256, 4), (256, 1)):
    x2 = torch.randn(rows, kk, device=dev, dtype=torch.bfloat16); dy2 = torch.randn(rows, nn, device=dev, dtype=torch.bfloat16)
    print('plain', kk, nn, timeit(lambda: dy2.t() @ x2))
    B=256; r=rows//B
    print('split', kk, nn, timeit(lambda: torch.bmm(dy2.view(B, r, nn).transpose(1,2), x2.view(B, r, kk)).sum(0, dtype=torch.float32)))
print('sum0 bf16', timeit(lambda: dy.sum(0)))
print('sum0 fp32 acc', timeit(lambda: dy.sum(0, dtype=torch.float32)))
ones = torch.ones(1, rows, device=dev, dtype=torch.bfloat16)
print('ones gemm', timeit(lambda: ones @ dy))
print('view sum', timeit(lambda: dy.view(256, rows//256, N).sum(1, dtype=torch.float32).sum(0)))
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev, dtype=torch.bfloat16)
print('linear', timeit(lambda: torch.nn.functional.linear(x, w, b)))
print('linear+relu', timeit(lambda: torch.relu(torch.nn.functional.linear(x, w, b))))
try:
    print('_addmm_activation relu', timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)))
except Exception as e: print('addmm_act failed', repr(e)[:200])
print('dx gemm', timeit(lambda: dy @ w))
print('relu bwd (threshold)', timeit(lambda: torch.ops.aten.threshold_backward(dy, x, 0)))
