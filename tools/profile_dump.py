import os, sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet
B = int(sys.argv[1])
u = synthetic_unet(W.sd15_unet_config(), max_batch=2*B, max_hw=(64, 64))
u.set_context(torch.randn(2*B, 77, 768))
x = torch.randn(2*B, 4, 64, 64, device='cuda'); s = torch.full((2*B,), 3.0, device='cuda')
for _ in range(2): u.forward(x, s)
torch.cuda.synchronize()
os.environ['LD_PROFILE_DUMP'] = '1'
p = u.profile(x, s)
print({k: round(v[0], 3) for k, v in p.items()})
