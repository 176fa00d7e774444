"""GPU parity of the sampling call surface: the reference's own sampler trajectories (goldens produced by its
common_ksampler / sample / KSAMPLER stack on the tiny UNet, fp32 CPU) against this package's mirror driving the HIP UNet.
Tolerance: a trajectory is 4-6 CFG steps (cfg 7-8 amplifies the cond/uncond difference) of an fp16-storage UNet whose
single-call bound is rel-L2 5e-3 — the end latent must agree to rel-L2 3e-2."""
import pytest
import torch

from conftest import load_golden, rel_l2
from lightdiffusion_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TRAJ_TOL = 3e-2


@pytest.fixture(scope="module")
def stack():
    from lightdiffusion_amd import nodes
    model, clip, vae = nodes.load_synthetic(DEV, max_batch=2, max_hw=(16, 16), tiny=True)
    return model, clip, vae


def conds(g):
    return [[g["pos"], {"pooled_output": None}]], [[g["neg"], {"pooled_output": None}]]


def test_euler_ancestral_txt2img_and_img2img(stack):
    from lightdiffusion_amd import nodes
    model = stack[0]
    g = load_golden("samplers")
    pos, neg = conds(g)
    lat = nodes.EmptyLatentImage().generate(128, 96, 1)[0]
    out = nodes.KSampler2().sample(model, 1234, 6, 7.5, "euler_ancestral", "normal", pos, neg, lat)[0]["samples"]
    assert out.device.type == "cpu" and out.shape == g["euler_a_txt2img"].shape
    assert rel_l2(out, g["euler_a_txt2img"]) < TRAJ_TOL
    out = nodes.KSampler2().sample(model, 77, 4, 8.0, "euler_ancestral", "normal", pos, neg, {"samples": g["lat2"]}, denoise=0.45)[0]["samples"]
    assert rel_l2(out, g["euler_a_img2img"]) < TRAJ_TOL


def test_dpmpp_2m_and_sde_with_injected_noise(stack):
    from lightdiffusion_amd import sampling as S
    model = stack[0]
    g = load_golden("samplers")
    pos, neg = conds(g)
    lat = torch.zeros(1, 4, 12, 16)
    sig = S.calculate_sigmas(model.get_model_object("model_sampling"), "karras", 6)
    noise = S.prepare_noise(lat, 99)
    out = S.sample(model, noise, pos, neg, 7.0, model.load_device, S.ksampler("dpmpp_2m_sde", {"eta": 0.0}), sig, model.model_options,
                   latent_image=lat, seed=99).cpu()
    assert rel_l2(out, g["dpmpp2m_eta0"]) < TRAJ_TOL
    gen = torch.Generator().manual_seed(5)
    ns = lambda s, sn: torch.randn(lat.shape, generator=gen).to(DEV)
    out = S.sample(model, noise, pos, neg, 7.0, model.load_device, S.ksampler("dpmpp_2m_sde", {"eta": 1.0, "noise_sampler": ns}), sig,
                   model.model_options, latent_image=lat, seed=99).cpu()
    assert rel_l2(out, g["dpmpp2m_sde_injected"]) < TRAJ_TOL
    # default eta=1 noise (torchsde stand-in): runs, finite, seed-reproducible
    a = S.sample(model, noise, pos, neg, 7.0, model.load_device, S.ksampler("dpmpp_2m_sde"), sig, model.model_options, latent_image=lat, seed=3)
    b = S.sample(model, noise, pos, neg, 7.0, model.load_device, S.ksampler("dpmpp_2m_sde"), sig, model.model_options, latent_image=lat, seed=3)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_toy_trajectories_on_device():
    """k-diffusion update arithmetic alone (toy denoiser), against the reference's functions."""
    from lightdiffusion_amd import sampling as S
    g = load_golden("samplers")
    toy = lambda x, s, **kw: x * (1.0 / (1.0 + s.view(-1, 1, 1, 1) ** 2))
    sig = S.get_sigmas_karras(8, 0.03, 14.6)
    x0 = g["toy_x0"].to(DEV)
    torch.manual_seed(7)
    assert rel_l2(S.sample_euler_ancestral(toy, x0, sig).cpu(), g["toy_euler_a"]) < 1e-5
    assert rel_l2(S.sample_dpmpp_2m_sde(toy, x0, sig, eta=0.0).cpu(), g["toy_dpmpp2m"]) < 1e-5
    assert rel_l2(S.sample_dpmpp_2m_sde(toy, x0, sig, eta=0.0, solver_type="heun").cpu(), g["toy_dpmpp2m_heun"]) < 1e-5


def test_dpm_adaptive(stack):
    """The GUI-default sampler (LD.py:10572-10576): toy denoiser exactly; tiny UNet through ksampler("dpm_adaptive").
    The PID controller makes accept/reject decisions on an error estimate, so fp16 rounding can move a step boundary:
    the end latent is held to the looser trajectory tolerance."""
    from lightdiffusion_amd import sampling as S
    g = load_golden("samplers")
    toy = lambda x, s, **kw: x * (1.0 / (1.0 + s.view(-1, 1, 1, 1) ** 2))
    x, info = S.sample_dpm_adaptive(toy, g["toy_x0"].to(DEV), 0.03, 14.6, return_info=True)
    assert [info["steps"], info["nfe"], info["n_accept"], info["n_reject"]] == g["toy_dpm_adaptive_info"].tolist()
    assert rel_l2(x.cpu(), g["toy_dpm_adaptive"]) < 1e-4
    model = stack[0]
    pos, neg = conds(g)
    lat = torch.zeros(1, 4, 12, 16)
    sig = S.calculate_sigmas(model.get_model_object("model_sampling"), "karras", 6)
    noise = S.prepare_noise(lat, 99)
    out = S.sample(model, noise, pos, neg, 7.0, model.load_device, S.ksampler("dpm_adaptive"), sig, model.model_options,
                   latent_image=lat, seed=99).cpu()
    assert rel_l2(out, g["dpm_adaptive_tiny"]) < TRAJ_TOL
    with pytest.raises(ValueError):
        S.sample_dpm_adaptive(toy, g["toy_x0"].to(DEV), 0.0, 14.6)


def test_graph_replay_equals_eager(stack):
    from lightdiffusion_amd.pipeline import CFGDenoiser
    unet = stack[0].model.diffusion_model
    g = load_golden("samplers")
    x = torch.randn(1, 4, 12, 16, generator=torch.Generator().manual_seed(1)).to(DEV) * 3
    outs = []
    for use_graph in (False, True):
        d = CFGDenoiser(unet, 1, 12, 16, 7.5, use_graph=use_graph)
        d.set_context(g["neg"], g["pos"])
        r = [d(x, 2.0).clone(), d(x, 0.5).clone(), d(x, 2.0).clone()]
        outs.append(r)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(outs[1][0], outs[1][2]) and not torch.equal(outs[1][0], outs[1][1])


def test_txt2img_end_to_end_tiny(stack):
    from lightdiffusion_amd import nodes
    model, clip, vae = stack
    toks = [[(49406, 1.0)] + [(1000 + i, 1.2 if i < 3 else 1.0) for i in range(10)] + [(49407, 1.0)] * 66]
    neg = [[(49406, 1.0)] + [(49407, 1.0)] * 76]
    img = nodes.txt2img(model, clip.clone(), vae, toks, neg, width=128, height=128, batch_size=2, seed=5, steps=4, cfg=6.0,
                        sampler_name="euler_ancestral", scheduler="normal")
    assert img.shape == (2, 128, 128, 3) and img.dtype == torch.float32 and img.device.type == "cpu"
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0 and torch.isfinite(img).all()
    img2 = nodes.txt2img(model, clip.clone(), vae, toks, neg, width=128, height=128, batch_size=2, seed=5, steps=4, cfg=6.0,
                         sampler_name="euler_ancestral", scheduler="normal")
    assert torch.equal(img, img2)


def test_checkpoint_loader_roundtrip(stack, tmp_path):
    """CheckpointLoaderSimple on a synthetic single-file safetensors == the directly constructed synthetic stack."""
    from safetensors.torch import save_file
    from lightdiffusion_amd import nodes
    from test_host_cpu import _synthetic_checkpoint
    sd, ucfg, vcfg, ccfg = _synthetic_checkpoint()
    path = str(tmp_path / "tiny_sd15.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    model, clip, vae = nodes.CheckpointLoaderSimple(DEV, max_batch=2, max_hw=(16, 16), clip_heads=4).load_checkpoint(path)
    g = load_golden("unet_tiny_16x16")
    a, b = model.model.diffusion_model, stack[0].model.diffusion_model
    for u in (a, b):
        u.set_context(g["ctx"])
    ya, yb = a.forward(g["x"].to(DEV), g["sigma"].to(DEV)), b.forward(g["x"].to(DEV), g["sigma"].to(DEV))
    assert torch.equal(ya, yb)
    z = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(3))
    assert torch.equal(vae.decode(z), stack[2].decode(z))


def test_multiple_conds_per_list_match_reference(stack):
    """Two positive and two negative entries (77 / 154 tokens; `area` / `strength` keys the reference's stripped
    get_area_and_mult ignores): per-list average of whole-latent predictions (calc_cond_batch, LD.py:2492-2591), 5 Euler-a
    steps on the tiny UNet against the reference's common_ksampler; the batch the hook sees is the reference's, row for row."""
    from lightdiffusion_amd import nodes
    model = stack[0]
    g = load_golden("multicond")
    pos = [[g["pos0"], {"pooled_output": None}], [g["pos1"], {"pooled_output": None, "area": (4, 4, 0, 0), "strength": 0.3}]]
    neg = [[g["neg0"], {"pooled_output": None}], [g["neg1"], {"pooled_output": None}]]
    seen = {}
    unet = model.model.diffusion_model

    class Hook:
        def __call__(self, apply_model, params):
            if not seen:
                seen.update(cou=list(params["cond_or_uncond"]), ctx=params["c"]["c_crossattn"].detach().cpu().clone())
            return unet(apply_model, params)

    m2 = model.clone()
    m2.set_model_unet_function_wrapper(Hook())
    lat = nodes.EmptyLatentImage().generate(128, 96, 2)[0]
    out = nodes.KSampler2().sample(m2, 4321, 5, 6.0, "euler_ancestral", "normal", pos, neg, lat)[0]["samples"]
    assert seen["cou"] == g["hook_cond_or_uncond"].tolist() and torch.equal(seen["ctx"], g["hook_ctx"])
    assert rel_l2(out, g["out"]) < TRAJ_TOL
    out2 = nodes.KSampler2().sample(model, 4321, 5, 6.0, "euler_ancestral", "normal", pos, neg, lat)[0]["samples"]   # wrapper = the UNet itself
    assert torch.equal(out, out2)


def test_rccl_group_and_graph_capture_in_one_process():
    """the "nccl" (= RCCL) backend at world = 1 on cuda:0, `dist.broadcast_conditioning`, then `KSampler2.sample`
    (captures + replays the step's hipGraph with the process group's watchdog thread alive: capture_error_mode="thread_local"), the
    wrapper hook's own graphs, then `dist.gather_images`; latents bitwise equal to the run without a process group.  A child process
    (tools/rccl_graph_check.py) so that a hung communicator cannot take the suite with it.  All a one-GPU box can show of config #4."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join("tools", "rccl_graph_check.py")], cwd=root, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rccl world-1 + hipGraph ok" in r.stdout
