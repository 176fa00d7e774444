import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, '/root/repo')
from lightdiffusion_amd import ops
which = sys.argv[1]
if which == "v4":
    x = torch.randn(16, 320, device="cuda", dtype=torch.float16); w = torch.randn(1280, 320, device="cuda", dtype=torch.float16)
elif which == "v3":
    x = torch.randn(65536, 320, device="cuda", dtype=torch.float16); w = torch.randn(320, 320, device="cuda", dtype=torch.float16)
else:
    x = torch.randn(65536, 1280, device="cuda", dtype=torch.float16); w = torch.randn(320, 1280, device="cuda", dtype=torch.float16)
for _ in range(3):
    y = ops.linear(x, w)
torch.cuda.synchronize()
print(which, "ok", float(y.float().abs().mean()))
