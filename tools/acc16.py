import os, sys, torch
sys.path.insert(0, "/root/repo")
from planerecnet_amd import ops
lib, _p, _s, check = ops.lib, ops._p, ops._stream, ops.check
def errs(y, w, x):
    wd, xd = w.double().cpu(), x.double().cpu()
    ref = torch.einsum("mk,bkp->bmp", wd, xd); mag = torch.einsum("mk,bkp->bmp", wd.abs(), xd.abs())
    e = (y.double().cpu() - ref) / (mag + 1e-300)
    return float(e.abs().max()), float(e.pow(2).mean().sqrt()), float(e.mean())
ops.set_split_gemm(mode=2)
print("piece format:", "fp16 x 2, %d products" % ops.split_products() if ops._POLICY["kind"] == 16 else "bf16 x 3, six products")
for (M, K, B, HW, dist) in [(1024,256,2,1200,"uniform"),(256,1024,2,1200,"uniform"),(200,72,3,1205,"uniform"),(512,2048,8,300,"uniform"),(1024,256,2,1200,"normal"),(1024,256,2,1200,"wide"),(256,256,2,4800,"relu")]:
    g = torch.Generator().manual_seed(M+K)
    if dist == "uniform":
        x = torch.rand(B,K,HW,generator=g)*2-1; w = (torch.rand(M,K,generator=g)*2-1)*K**-0.5
    elif dist == "normal":
        x = torch.randn(B,K,HW,generator=g); w = torch.randn(M,K,generator=g)*K**-0.5
    elif dist == "relu":
        x = torch.relu(torch.randn(B,K,HW,generator=g)); w = torch.randn(M,K,generator=g)*K**-0.5
    else:   # wide dynamic range: log-uniform magnitudes over 12 decades, rows / columns with different scales
        x = torch.randn(B,K,HW,generator=g) * torch.exp(torch.randn(B,K,HW,generator=g)*3) * torch.exp(torch.randn(B,1,HW,generator=g)*6)
        w = torch.randn(M,K,generator=g) * torch.exp(torch.randn(M,K,generator=g)*3) * torch.exp(torch.randn(M,1,generator=g)*6)
    x, w = x.cuda(), w.cuda()
    out = {}
    for mode in (2, 0):
        ops.set_split_gemm(mode=mode)
        _, ref, nb, _, _ = ops._desc(B, K, 1, HW, M, 1, 1, 0, 1, HW, 0, 1, 0)
        ws = torch.empty(max(nb,16)//4, device="cuda"); y = torch.full((B,M,HW), float("nan"), device="cuda")
        check(lib.prn_conv2d_fwd(ref, _p(x), _p(w), None, None, _p(y), _p(ws), _s()), "conv"); torch.cuda.synchronize()
        assert bool(torch.isfinite(y).all())
        out[mode] = errs(y, w, x)
    print("%-28s split max %.2e rms %.2e mean %+.1e | fp32 max %.2e rms %.2e mean %+.1e" % ((M,K,B,HW,dist).__str__(), *out[2], *out[0]))
