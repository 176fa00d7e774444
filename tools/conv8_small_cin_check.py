"""Short-loop shapes through ld_op_conv (conv8 with 2 .. 14 sub-slabs per workgroup, odd and even; the general kernels elsewhere) against torch fp32."""
import math, sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from lightdiffusion_amd import ops
def r16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(shape, generator=g) * scale).half()
bad = 0
for hw in (8, 16, 32, 64):
    for cin in (32, 64, 96, 128, 160, 224):
        for cout in (80, 160):
            x = r16((2, cin, hw, hw), 1); w = r16((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin)); b = r16((cout,), 3, 0.1)
            ref = F.conv2d(x.float().cuda(), w.float().cuda(), b.float().cuda(), padding=1)
            try:
                y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().cuda(), ops.repack_conv_weight(w.cuda()), b.cuda())
            except Exception as e:   # a shape no kernel takes (channel counts off the MFMA path's multiples): declined loudly, not wrong
                print("declined", hw, cin, cout, str(e)[:60]); continue
            err = float((y.float().permute(0, 3, 1, 2) - ref).norm() / ref.norm())
            if not (err < 2e-3): bad += 1; print("BAD", hw, cin, cout, err)
print("small-Cin conv sweep:", "ok" if bad == 0 else f"{bad} bad")
