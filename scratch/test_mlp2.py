import torch, time
dev='cuda'
rows = 1<<20
def timeit(f, n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
bf=torch.bfloat16
x20 = torch.randn(rows, 20, device=dev, dtype=bf); x32 = torch.zeros(rows, 32, device=dev, dtype=bf)
w20 = torch.randn(256, 20, device=dev, dtype=bf); w32 = torch.zeros(256, 32, device=dev, dtype=bf); b=torch.randn(256, device=dev, dtype=bf)
print('first layer K=20', timeit(lambda: torch._addmm_activation(b, x20, w20.t())))
print('first layer K=32', timeit(lambda: torch._addmm_activation(b, x32, w32.t())))
h = torch.randn(rows, 256, device=dev, dtype=bf)
for n in (4, 1, 8, 16):
    w = torch.randn(n, 256, device=dev, dtype=bf); bb = torch.randn(n, device=dev, dtype=bf)
    print('last layer N=%d'%n, timeit(lambda: torch.nn.functional.linear(h, w, bb)))
    try: print('  fp32 out', timeit(lambda: torch.mm(h, w.t(), out_dtype=torch.float32)))
    except Exception as e: print('  out_dtype fail', repr(e)[:100])
    dy = torch.randn(rows, n, device=dev, dtype=bf)
    print('  dx = dy W', timeit(lambda: dy @ w))
    B=128; r=rows//B
    print('  dW split', timeit(lambda: torch.bmm(dy.view(B, r, n).transpose(1,2), h.view(B, r, 256)).sum(0, dtype=torch.float32)))
    if n==1:
        print('  dW mv', timeit(lambda: torch.mv(h.t(), dy.view(-1))))
        print('  dW bmm swapped', timeit(lambda: torch.bmm(h.view(B, r, 256).transpose(1,2), dy.view(B, r, 1)).sum(0, dtype=torch.float32)))
        dyp = torch.zeros(rows, 8, device=dev, dtype=bf)
        print('  dW padded8', timeit(lambda: torch.bmm(dyp.view(B, r, 8).transpose(1,2), h.view(B, r, 256)).sum(0, dtype=torch.float32)))
# fp32 versions
xf = torch.randn(rows, 256, device=dev); wf = torch.randn(256,256, device=dev); bfl = torch.randn(256, device=dev)
print('fp32 linear+relu fused', timeit(lambda: torch._addmm_activation(bfl, xf, wf.t())))
print('fp32 dW plain', timeit(lambda: xf.t() @ xf))
B=128; r=rows//B
print('fp32 dW split', timeit(lambda: torch.bmm(xf.view(B, r, 256).transpose(1,2), xf.view(B, r, 256)).sum(0)))
