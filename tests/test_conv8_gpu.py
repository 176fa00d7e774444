"""GPU parity of the row-resident convolution (csrc/conv8.hip) — the two-image (CFG pair) levels of a batch-1 step: ResBlock1's
GroupNorm + SiLU (norm.hip) + conv3x3 (+ time-embedding row + skip, LD.py:5224-5287) and Upsample1's nearest-2x + conv (LD.py:5141-5152),
called through the C ABI (ld_op_groupnorm_conv / ld_op_conv) against a torch fp32 evaluation of the same op on the same fp16-rounded
inputs.  Cases with more than two images take the general kernels through the same entry points (conv8_plan declines them).
Tolerance: rel-L2 <= 2e-3 (fp16 storage, fp32 accumulate; the channel-slab partial sums are fp32 and summed in a fixed order)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-3


@pytest.fixture(scope="module")
def ops():
    from lightdiffusion_amd import ops as o
    from lightdiffusion_amd._lib import lib
    lib()
    return o


def r16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# (h = w, c1, c2, cout, time-embedding row, residual): every ResBlock convolution of SD1.5's 16x16 / 8x8 levels and the middle block at
# UNet batch 2 — in_layers of the down path (640 -> 1280, 1280 -> 1280), of the up path (concat 1280 + 1280 / 1280 + 640: groups of 80 / 60
# channels, the second one straddling the 16-channel sub-slabs AND the source boundary), out_layers (residual = skip)
# and the 32x32 / 64x64 levels (patches of four / two image rows; at 64x64 one workgroup sweeps all of K), plus batch 16 / 4 / 1 cases
@pytest.mark.parametrize("n,hw,c1,c2,cout,rv,res", [
    (2, 16, 1280, 0, 1280, True, False), (2, 16, 1280, 0, 1280, False, True), (2, 16, 640, 0, 1280, True, False),
    (2, 16, 1280, 1280, 1280, True, False), (2, 16, 1280, 640, 1280, True, True),
    (2, 8, 1280, 0, 1280, True, True), (2, 8, 1280, 1280, 1280, True, False), (2, 8, 1280, 0, 640, False, False),
    (2, 32, 640, 0, 640, True, True), (2, 32, 320, 0, 640, True, False), (2, 32, 640, 320, 640, True, False), (2, 32, 1280, 640, 640, False, True),
    (2, 64, 320, 0, 320, True, True), (2, 64, 320, 320, 320, True, False), (2, 64, 640, 320, 320, False, False),
    (16, 8, 1280, 0, 1280, True, True), (4, 16, 1280, 0, 1280, True, False), (1, 16, 640, 0, 640, False, False)])
def test_groupnorm_silu_conv_row_resident(ops, n, hw, c1, c2, cout, rv, res):
    h, w = hw, hw
    x1 = r16((n, c1, h, w), 201, 2.0) + 0.5
    x2 = (r16((n, c2, h, w), 202) - 1.0) if c2 else None
    cin = c1 + c2
    ga, be = (1 + 0.1 * r16((cin,), 203).float()).half(), r16((cin,), 204, 0.1)
    wt, b = r16((cout, cin, 3, 3), 205, 1 / math.sqrt(9 * cin)), r16((cout,), 206, 0.1)
    xin = x1.float().to(DEV) if x2 is None else torch.cat([x1.float(), x2.float()], 1).to(DEV)
    g = F.silu(F.group_norm(xin, 32, ga.float().to(DEV), be.float().to(DEV), 1e-5))
    ref = F.conv2d(g, wt.float().to(DEV), b.float().to(DEV), padding=1)
    rowvec = r16((n, cout), 207) if rv else None
    if rv:
        ref = ref + rowvec.float().to(DEV)[:, :, None, None]
    r = r16(tuple(ref.shape), 208) if res else None
    if res:
        ref = ref + r.float().to(DEV)
    wp = ops.repack_conv_weight(wt.to(DEV))
    args = (nhwc(x1).to(DEV), ga.to(DEV), be.to(DEV), 1e-5, wp, b.to(DEV), None if x2 is None else nhwc(x2).to(DEV),
            None if rowvec is None else rowvec.to(DEV), None if r is None else nhwc(r).to(DEV))
    y = ops.group_norm_silu_conv2d(*args)
    assert rel_l2(nchw(y.float().cpu()), ref.cpu()) < TOL
    # the in-launch reduction sums the slabs in a fixed order: replays are bitwise identical, and the counters it leaves behind are zero
    # (a second call on the same scratch would otherwise hang or reduce early)
    for _ in range(3):
        y2 = ops.group_norm_silu_conv2d(*args)
        assert torch.equal(y, y2)


@pytest.mark.parametrize("n,hw,cin,cout,up,rv,res", [
    (2, 16, 1280, 1280, True, False, False),     # Upsample1 of the 8x8 level: nearest 2x inside the halo loader
    (2, 32, 1280, 1280, True, False, False),     # ... of the 16x16 level
    (2, 64, 640, 640, True, False, False),       # ... of the 32x32 level
    (2, 16, 1280, 1280, False, True, True),      # plain (no GroupNorm) route
    (2, 8, 1280, 1280, False, False, True),
    (2, 64, 320, 320, False, True, False),
    (2, 32, 640, 640, False, True, True)])       # 128 (patch, N tile) counters of the in-launch reduction
def test_conv_row_resident(ops, n, hw, cin, cout, up, rv, res):
    hs = hw // 2 if up else hw
    x = r16((n, cin, hs, hs), 211)
    wt, b = r16((cout, cin, 3, 3), 212, 1 / math.sqrt(9 * cin)), r16((cout,), 213, 0.1)
    xin = x.float().to(DEV)
    if up:
        xin = F.interpolate(xin, size=(hw, hw), mode="nearest")
    ref = F.conv2d(xin, wt.float().to(DEV), b.float().to(DEV), padding=1)
    rowvec = r16((n, cout), 214) if rv else None
    if rv:
        ref = ref + rowvec.float().to(DEV)[:, :, None, None]
    r = r16(tuple(ref.shape), 215) if res else None
    if res:
        ref = ref + r.float().to(DEV)
    y = ops.conv2d(nhwc(x).to(DEV), ops.repack_conv_weight(wt.to(DEV)), b.to(DEV), 3, 1, None, (hw, hw) if up else None,
                   None if rowvec is None else rowvec.to(DEV), None if r is None else nhwc(r).to(DEV))
    assert rel_l2(nchw(y.float().cpu()), ref.cpu()) < TOL


# GroupNorm partial statistics written by the PRODUCING convolution (so the GroupNorm that follows runs one pass): every kernel that writes
# them — the halo convolution's generic epilogue at 256- and 128-column tiles (the VAE decoder's stages, LD.py:3560-3576; also behind the
# nearest-2x upsampling), the row-resident kernel (two-image levels of the UNet) and the split-K second pass (8 x 8 level at UNet batch 16).
# Checked against the statistics of the tensor the kernel stored: per (image, group) sum and sum of squares over all chunks, in fp64.
@pytest.mark.parametrize("n,hw,cin,cout,up,res,expect", [
    (8, 64, 512, 512, False, True, True),        # conv6 <W64, 256-column tiles>: 16-channel groups = two 8-channel chunks
    (4, 128, 256, 256, False, False, True),      # conv6 <W128, 256>: 8-channel groups
    (2, 256, 128, 128, False, True, True),       # conv6 <W128, 128 x 512 tiles>: 4-channel groups, two per chunk; image cut into 128-pixel bands
    (4, 128, 256, 256, True, False, True),       # ... behind the nearest-2x upsampling (64 -> 128)
    (16, 64, 320, 320, False, True, None),       # conv6 <W64, 320-column tile> (the UNet's level 0 at batch 16; 10-channel groups): no partials from its epilogue today
    (4, 128, 640, 320, False, False, None),      # (round 6 built them — tools/experiments/gn_partials_v5_epilogue_n320_r06.patch.txt — and measured +-0 per forward); either way exact
    (2, 16, 1280, 1280, False, True, True),      # conv8: 16-pixel chunks
    (2, 64, 320, 320, False, False, True),       # conv8 at 64 x 64
    (16, 8, 1280, 1280, False, True, True),      # 128 x 160 kernel + split-K reduce with GroupNorm partials
    (2, 16, 960, 960, False, True, None),        # groups of 30 channels straddle conv8's 80-column tiles: it must decline (or be exact) — either
    (2, 16, 640, 1920, False, False, None)])     # route may or may not write partials, but what it writes must be right.  Groups of 60.
def test_conv_writes_groupnorm_partials_of_its_output(ops, n, hw, cin, cout, up, res, expect):
    hs = hw // 2 if up else hw
    x = r16((n, cin, hs, hs), 221)
    wt, b = r16((cout, cin, 3, 3), 222, 1 / math.sqrt(9 * cin)), r16((cout,), 223, 0.1)
    r = r16((n, cout, hw, hw), 224) if res else None
    y, part = ops.conv2d_gn_partials(nhwc(x).to(DEV), ops.repack_conv_weight(wt.to(DEV)), b.to(DEV), (hw, hw) if up else None,
                                     None if r is None else nhwc(r).to(DEV))
    xin = x.float().to(DEV)
    if up:
        xin = F.interpolate(xin, size=(hw, hw), mode="nearest")
    ref = F.conv2d(xin, wt.float().to(DEV), b.float().to(DEV), padding=1)
    if res:
        ref = ref + r.float().to(DEV)
    assert rel_l2(nchw(y.float().cpu()), ref.cpu()) < TOL
    if expect is None and part is None:
        return
    assert (part is not None) == (expect is None or expect)
    got = part.double().sum(1)                                          # [n, 32, 2]
    yg = y.double().view(n, hw * hw, 32, cout // 32)                     # the STORED fp16 values, grouped
    want = torch.stack([yg.sum((1, 3)), (yg * yg).sum((1, 3))], -1)
    assert float((got[..., 0] - want[..., 0]).abs().max()) < 1e-3 * float(yg.abs().sum((1, 3)).max())
    assert float(((got[..., 1] - want[..., 1]).abs() / want[..., 1]).max()) < 1e-4


def test_lone_reducer_route_is_bitwise_the_normal_one():
    """conv8's in-launch reduction when a workgroup's bounded wait for its peers runs out (conv8.hip: the last arriver sums the whole tile
    alone): forced in the A/B build with LD_C8_NO_WAIT=1 and compared bitwise (outputs + GroupNorm partials, five shapes, three replays
    each) with the normal route.  Runs tools/conv8_timeout_check.py, which starts one fresh child process per mode with
    LD_MI355X_LIB = the A/B library (never a re-exec of a process that has touched the GPU)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = os.path.join(root, "lightdiffusion_amd", "libld_mi355x_ab.so")
    if not os.path.exists(ab):
        pytest.skip("A/B library not built (make -C lightdiffusion_amd/csrc ab)")
    env = {k: v for k, v in os.environ.items() if k not in ("LD_MI355X_LIB", "LD_C8_NO_WAIT")}
    r = subprocess.run([sys.executable, os.path.join("tools", "conv8_timeout_check.py")], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical" in r.stdout
