"""GPU: edge cases of the hot path — ragged / odd shapes, long prompts, maximum reserved sizes, and the error statuses of
the C ABI.  References are the pinned oracle (tests/test_oracle_vs_golden.py) on the same inputs."""
import pytest
import torch

from conftest import rel_l2
from lightdiffusion_amd import weights as W
from oracle import sd15_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def unet():
    from lightdiffusion_amd.unet import synthetic_unet
    return synthetic_unet(W.tiny_unet_config(), max_batch=4, max_hw=(16, 16), max_tokens=154)


@pytest.fixture(scope="module")
def sd():
    return W.synth_state_dict(W.unet_param_shapes(W.tiny_unet_config()))


def ref_denoised(sd, x, sigma, ctx):
    return O.apply_model(sd, W.tiny_unet_config(), O.ModelSampling(), x, sigma, ctx)


@pytest.mark.parametrize("h,w,n", [(10, 6, 2), (7, 9, 1), (16, 16, 4), (1, 8, 2), (13, 4, 3)])
def test_odd_and_ragged_latents(unet, sd, h, w, n):
    """odd sizes exercise: stride-2 convs with ragged edges, nearest-resize to a non-2x skip size (Upsample1 output_shape,
    LD.py:5141-5149), attention with token counts that are not multiples of 8 / 32 / 64, batch 1..4."""
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(n, 4, h, w, generator=g) * 2.0
    sigma = torch.rand(n, generator=g) * 5 + 0.05
    ctx = torch.randn(n, 77, 64, generator=g)
    unet.set_context(ctx)
    y = unet.forward(x.to(DEV), sigma.to(DEV)).cpu()
    assert torch.isfinite(y).all()
    assert rel_l2(y, ref_denoised(sd, x, sigma, ctx)) < 5e-3


def test_long_prompt_two_chunks(unet, sd):
    """> 75 tokens: the reference concatenates 77-token chunks -> 154 context tokens (LD.py:4540-4569)."""
    g = torch.Generator().manual_seed(5)
    x, sigma, ctx = torch.randn(2, 4, 8, 8, generator=g), torch.tensor([3.0, 0.4]), torch.randn(2, 154, 64, generator=g)
    unet.set_context(ctx)
    y = unet.forward(x.to(DEV), sigma.to(DEV)).cpu()
    assert rel_l2(y, ref_denoised(sd, x, sigma, ctx)) < 5e-3


def test_sigma_extremes(unet, sd):
    """sigma_min / sigma_max of the table and beyond: the timestep argmin saturates at 0 / 999."""
    ms = O.ModelSampling()
    g = torch.Generator().manual_seed(6)
    ctx = torch.randn(4, 77, 64, generator=g)
    sigma = torch.tensor([float(ms.sigma_min), float(ms.sigma_max), 1e-3, 80.0])
    x = torch.randn(4, 4, 8, 8, generator=g) * sigma.view(-1, 1, 1, 1)
    unet.set_context(ctx)
    y = unet.forward(x.to(DEV), sigma.to(DEV)).cpu()
    assert rel_l2(y, ref_denoised(sd, x, sigma, ctx)) < 5e-3


def test_error_statuses(unet):
    from lightdiffusion_amd._lib import ERR_SHAPE, ERR_STATE, LDError
    from lightdiffusion_amd.unet import synthetic_unet
    x, s = torch.randn(2, 4, 8, 8, device=DEV), torch.ones(2, device=DEV)
    fresh = synthetic_unet(W.tiny_unet_config(), max_batch=2, max_hw=(8, 8))
    with pytest.raises(LDError) as e:                       # forward before set_context
        fresh.forward(x, s)
    assert e.value.status == ERR_STATE
    fresh.set_context(torch.randn(2, 77, 64))
    with pytest.raises(LDError) as e:                       # batch does not match the context batch
        fresh.forward(torch.randn(1, 4, 8, 8, device=DEV), torch.ones(1, device=DEV))
    assert e.value.status == ERR_SHAPE
    with pytest.raises(LDError) as e:                       # larger than the reserved workspace
        fresh.forward(torch.randn(2, 4, 32, 32, device=DEV), s)
    assert e.value.status == ERR_SHAPE
    import ctypes as C
    from lightdiffusion_amd._lib import F32, lib
    big = torch.randn(2, 154, 64, device=DEV)
    # the C ABI refuses more context tokens / samples than reserved ...
    assert lib().ld_unet_set_context(fresh._h, big.data_ptr(), F32, 2, 154, torch.cuda.current_stream().cuda_stream) == ERR_SHAPE
    assert lib().ld_unet_set_context(fresh._h, big.data_ptr(), F32, 3, 77, torch.cuda.current_stream().cuda_stream) == ERR_SHAPE
    y = fresh.forward(x, s)                                 # the handle is still usable after refused calls
    assert torch.isfinite(y).all()
    # ... while the host wrapper grows the plan instead, like the reference (any number of 77-token chunks, any batch)
    epoch = fresh.reserve_epoch
    fresh.set_context(torch.randn(3, 154, 64))
    assert fresh.reserve_epoch == epoch + 1
    assert torch.isfinite(fresh.forward(torch.randn(3, 4, 8, 8, device=DEV), torch.ones(3, device=DEV))).all()
    with pytest.raises(KeyError):                           # a checkpoint that lacks a parameter is rejected at load
        from lightdiffusion_amd.unet import MI355XUNet
        MI355XUNet(W.tiny_unet_config(), {"input_blocks.0.0.weight": torch.zeros(64, 4, 3, 3)})


def test_op_shape_errors():
    from lightdiffusion_amd import ops
    from lightdiffusion_amd._lib import ERR_SHAPE, LDError
    with pytest.raises(LDError) as e:                       # conv channels must be multiples of 64 on the MFMA path
        ops.conv2d(torch.zeros(1, 4, 4, 48, device=DEV, dtype=torch.float16), torch.zeros(64, 9 * 48, device=DEV, dtype=torch.float16), None)
    assert e.value.status == ERR_SHAPE
    with pytest.raises(LDError) as e:                       # head dim must be a multiple of 8
        ops.attention(torch.zeros(1, 8, 20, device=DEV, dtype=torch.float16), torch.zeros(1, 8, 20, device=DEV, dtype=torch.float16),
                      torch.zeros(1, 8, 20, device=DEV, dtype=torch.float16), 2)
    assert e.value.status == ERR_SHAPE
    with pytest.raises(LDError):                            # GroupNorm needs C % 32 == 0
        ops.group_norm(torch.zeros(1, 4, 40, device=DEV, dtype=torch.float16), torch.ones(40, device=DEV, dtype=torch.float16),
                       torch.zeros(40, device=DEV, dtype=torch.float16), 1e-5)


def test_vae_odd_and_batch():
    from lightdiffusion_amd.unet import synthetic_vae
    cfg = W.tiny_vae_config()
    sdv = W.synth_state_dict(W.vae_decoder_param_shapes(cfg))
    v = synthetic_vae(cfg, max_batch=3, max_hw=(8, 8))
    g = torch.Generator().manual_seed(9)
    for shape in ((3, 4, 8, 8), (2, 4, 4, 6), (1, 4, 2, 4)):
        z = torch.randn(shape, generator=g)
        img = v.decode(z)
        ref = O.vae_decode(sdv, cfg, z)
        assert img.shape == ref.shape and float((img - ref).abs().max()) < 2.0 / 255.0


def test_vae_decode_splits_the_batch_by_memory():
    """VAE.decode decodes the batch in slices that fit the free memory (LD.py:6357-6381).  With a budget that holds one image of a batch of
    three the decode runs as three slices and gives the whole-batch result (rounding follows the batch: tile choices); the plan query itself
    grows with the batch and allocates nothing."""
    from lightdiffusion_amd import weights as W
    from lightdiffusion_amd._lib import lib
    from lightdiffusion_amd.unet import synthetic_vae
    cfg = W.tiny_vae_config()
    z = torch.randn(3, 4, 16, 24, generator=torch.Generator().manual_seed(8))
    whole = synthetic_vae(cfg, max_batch=3, max_hw=(16, 24)).decode(z)
    vae = synthetic_vae(cfg, max_batch=1, max_hw=(8, 8))
    ws0 = vae.workspace_bytes
    need1, need3 = (lib().ld_vae_plan_bytes(vae._h, n, 16, 24) for n in (1, 3))
    assert 0 < need1 < need3 and vae.workspace_bytes == ws0
    vae.memory_budget = need1 + 1024
    assert vae.batch_number(3, 16, 24) == 1
    sliced = vae.decode(z)
    assert vae._reserved[0] == 1 and sliced.shape == whole.shape
    assert float((sliced - whole).abs().max()) < 2.0 / 255.0
    vae.memory_budget = None                               # the real free memory holds all three
    assert vae.batch_number(3, 16, 24) == 3
