"""Full-length, full-size end-to-end parity (SURVEY §8 row g1 / c "tolerance anchor"): the product call surface
(token ids -> CLIPTextModelHIP -> KSampler2 / sampling.sample -> hipGraph CFG denoiser -> VAEDecode) on the SD1.5-size
synthetic net against the REFERENCE's own call stack run fp32 on the CPU (oracle/make_golden.py e2e):

  config #2  B=1, 20 steps dpmpp_2m_sde eta=0 ("DPM++ 2M") / karras, cfg 7, seed 2002, 512^2 decode
  config #3  B=2, 30 steps euler_ancestral / normal, cfg 7, seed 3003 (host-generator noise) — also as rows 0-1 of a batch of 8
  config #5  bislerp x2 of #2's latent -> 10 euler_ancestral steps at denoise 0.45, cfg 8, seed 5005 -> 1024^2 decode

Every run records the per-step rel-L2 of the latent against the reference's trajectory and the final image error in /255
into gpurun_out/e2e_parity.json (the numbers DESIGN.md §2 quotes).  The yardstick is the fixture's `anchor_*`: the SAME
reference run with its UNet weights rounded to fp16 (what the reference's own unet_dtype1 stores) against its fp32 self."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = {}

# Stated bounds.  Measured on the MI355X (round 3; every run rewrites gpurun_out/e2e_parity.json):
#   config #2  final latent rel-L2 2.1e-3 (every step <= 2.2e-3; the reference against its own fp16-weight self: 8.8e-4),
#              image max-abs 1.03 / 255, mean-abs 0.15 / 255
#   config #3  2.4e-3 at batch 2 and as rows of a batch of 8; image max-abs 1.30 / 255, mean-abs 0.17 / 255
#   config #5  1.7e-4 (10 steps from sigma 1.28 on a latent that starts as the reference's); image max-abs 0.65 / 255
# The synthetic-weight net is NOT chaotic at cfg 7 / 8: the error is set in the first steps and stays flat, so no cfg-1 variant is needed.
LATENT_TOL = {"cfg2": 5e-3, "cfg3": 5e-3, "cfg5": 1e-3}         # rel-L2 of the final latent AND of the latent entering every step
IMG_MAX_TOL = 2.5                                                # max |diff| in /255 over the subsampled image and the 96x96 crop
IMG_MEAN_TOL = 0.4                                               # mean |diff| in /255


def _dump():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "e2e_parity.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


@pytest.fixture(scope="module")
def stack():
    from lightdiffusion_amd import nodes
    model, clip, vae = nodes.load_synthetic(DEV, max_batch=2, max_hw=(64, 64))
    clip = nodes.CLIPSetLastLayer().set_last_layer(clip, -2)[0]            # LD.py:10033-10035
    return model, clip, vae


@pytest.fixture(scope="module")
def conds(stack):
    """positive / negative conditioning from the fixture's token ids through the HIP text model."""
    g = load_golden("e2e_cond")
    _, clip, _ = stack
    pairs = lambda ids, w: [[(int(t), float(x)) for t, x in zip(ids.tolist(), w.tolist())]]
    cpos, ppos = clip.encode_from_tokens(pairs(g["pos_ids"], g["pos_w"]), return_pooled=True)
    cneg, pneg = clip.encode_from_tokens(pairs(g["neg_ids"], torch.ones(77)), return_pooled=True)
    REPORT["clip"] = {"cond_pos_rel_l2": rel_l2(cpos, g["cond_pos"]), "cond_neg_rel_l2": rel_l2(cneg, g["cond_neg"])}
    _dump()
    return [[cpos, {"pooled_output": ppos}]], [[cneg, {"pooled_output": pneg}]], g


def test_clip_conditioning_from_token_ids(conds):
    """weighted prompt (3 weights != 1: the empty-prompt lerp path) and the empty negative prompt, clip skip -2, against the
    reference's SDClipModel.encode_token_weights (LD.py:4540-4569, 4692-4724)."""
    pos, neg, g = conds
    assert pos[0][0].shape == (1, 77, 768) and pos[0][0].device.type == "cpu"
    assert rel_l2(pos[0][0], g["cond_pos"]) < 5e-3 and rel_l2(neg[0][0], g["cond_neg"]) < 5e-3


class _Traj:
    """sampler callback: the latent after every step, row 0, every second pixel (what the fixture's recorder kept)."""

    def __init__(self):
        self.xs = []

    def __call__(self, d):
        self.xs.append(d["x"][0:1, :, ::2, ::2].float().cpu().clone())


def _curve(traj, g_traj):
    """rel-L2 of the latent entering step i (i >= 1) against the reference's; entry 0 is the scaled initial noise."""
    n = min(len(traj), g_traj.shape[0] - 1)
    return [rel_l2(traj[i], g_traj[i + 1:i + 2]) for i in range(n)]


def _image_report(img, g, prefix=""):
    sub = g[prefix + "img_sub"]
    step = img.shape[1] // sub.shape[1]
    d_sub = (img[:, ::step, ::step] - sub).abs()
    c0 = int(g[prefix + "crop_at"][0])
    d_crop = (img[:, c0:c0 + 96, c0:c0 + 96] - g[prefix + "crop"]).abs()
    return {"max_abs_255": 255 * float(max(d_sub.max(), d_crop.max())), "mean_abs_255": 255 * float(d_sub.mean()),
            "crop_mean_abs_255": 255 * float(d_crop.mean()), "p99_abs_255": 255 * float(torch.quantile(d_sub.flatten(), 0.99)),
            "frac_gt_2": float((d_sub > 2 / 255).float().mean()), "ref_saturated": float(g[prefix + "img_saturated"]),
            "mean_err": abs(float(img.mean()) - float(g[prefix + "img_mean"]))}


def test_config2_dpmpp2m_20_steps_and_decode(stack, conds):
    from lightdiffusion_amd import nodes
    from lightdiffusion_amd import sampling as S
    model, _, vae = stack
    pos, neg, _ = conds
    g = load_golden("e2e_cfg2")
    lat = nodes.EmptyLatentImage().generate(512, 512, 1)[0]["samples"]
    sig = S.calculate_sigmas(model.get_model_object("model_sampling"), "karras", 20)
    noise = S.prepare_noise(lat, 2002)
    tr = _Traj()
    out = S.sample(model, noise, pos, neg, float(g["cfg"]), model.load_device, S.ksampler("dpmpp_2m_sde", {"eta": 0.0}), sig,
                   model.model_options, latent_image=lat, callback=tr, seed=2002).cpu()
    curve = _curve(tr.xs, g["traj_sub"])
    img = nodes.VAEDecode().decode(vae, {"samples": out})[0]
    rep = {"latent_rel_l2": rel_l2(out, g["latent"]), "traj_rel_l2": curve, "image": _image_report(img, g),
           "anchor_latent_rel_l2": float(g["anchor_latent_rel"]), "anchor_traj_rel_l2": g["anchor_traj_rel"].tolist(),
           "anchor_img_max_abs_255": 255 * float(g["anchor_img_maxabs"]), "anchor_img_mean_abs_255": 255 * float(g["anchor_img_meanabs"])}
    # the same run from the REFERENCE's conditioning (separates the text-model error from the sampler / UNet drift)
    gc = load_golden("e2e_cond")
    out_rc = S.sample(model, noise, [[gc["cond_pos"], {}]], [[gc["cond_neg"], {}]], float(g["cfg"]), model.load_device,
                      S.ksampler("dpmpp_2m_sde", {"eta": 0.0}), sig, model.model_options, latent_image=lat, seed=2002).cpu()
    rep["latent_rel_l2_reference_cond"] = rel_l2(out_rc, g["latent"])
    REPORT["cfg2"] = rep
    _dump()
    assert torch.isfinite(out).all() and out.shape == g["latent"].shape
    assert rep["latent_rel_l2"] < LATENT_TOL["cfg2"] and max(curve) < LATENT_TOL["cfg2"], rep
    assert rep["latent_rel_l2_reference_cond"] < LATENT_TOL["cfg2"], rep
    assert rep["image"]["max_abs_255"] < IMG_MAX_TOL and rep["image"]["mean_abs_255"] < IMG_MEAN_TOL, rep


def test_config3_euler_a_30_steps_batch2_and_as_rows_of_batch8(stack, conds):
    """KSampler2 at B=2 (the global host generator supplies initial and per-step noise in the reference's order), then the same
    two images as rows 0-1 of a batch of 8 (rows 2-7 other noise): batch-16 tile / split choices, same images."""
    from lightdiffusion_amd import nodes
    from lightdiffusion_amd import sampling as S
    model, _, vae = stack
    pos, neg, _ = conds
    g = load_golden("e2e_cfg3")
    lat = nodes.EmptyLatentImage().generate(512, 512, 2)[0]
    out = nodes.KSampler2().sample(model, 3003, 30, float(g["cfg"]), "euler_ancestral", "normal", pos, neg, lat)[0]["samples"]
    img = nodes.VAEDecode().decode(vae, {"samples": out})[0]
    rep = {"latent_rel_l2": rel_l2(out, g["latent"]), "row_rel_l2": [rel_l2(out[i], g["latent"][i]) for i in range(2)],
           "image_r0": _image_report(img[0:1], g, "r0_"), "image_r1": _image_report(img[1:2], g, "r1_")}
    # trajectory of row 0 + the batch-8 variant through the sampler-level entry with an injected noise sampler
    sig = S.calculate_sigmas(model.get_model_object("model_sampling"), "normal", 30)
    other = torch.Generator().manual_seed(99)

    def batch8_noise_sampler(sigma, sigma_next):
        ref_rows = torch.randn(2, 4, 64, 64)                                  # the reference's draw: global generator, batch 2
        return torch.cat([ref_rows, torch.randn(6, 4, 64, 64, generator=other)]).to(DEV)

    noise8 = torch.cat([S.prepare_noise(torch.zeros(2, 4, 64, 64), 3003), torch.randn(6, 4, 64, 64, generator=other)])
    tr = _Traj()
    out8 = S.sample(model, noise8, pos, neg, float(g["cfg"]), model.load_device,
                    S.ksampler("euler_ancestral", {"noise_sampler": batch8_noise_sampler}), sig, model.model_options,
                    latent_image=torch.zeros(8, 4, 64, 64), callback=tr, seed=3003).cpu()
    rep["batch8_rows01_latent_rel_l2"] = rel_l2(out8[:2], g["latent"])
    rep["batch8_traj_rel_l2"] = _curve(tr.xs, g["traj_sub"])
    rep["batch8_vs_batch2_rel_l2"] = rel_l2(out8[:2], out)
    REPORT["cfg3"] = rep
    _dump()
    assert torch.isfinite(out8).all()
    assert rep["latent_rel_l2"] < LATENT_TOL["cfg3"] and rep["batch8_rows01_latent_rel_l2"] < LATENT_TOL["cfg3"], rep
    assert max(rep["batch8_traj_rel_l2"]) < LATENT_TOL["cfg3"], rep
    for k in ("image_r0", "image_r1"):
        assert rep[k]["max_abs_255"] < IMG_MAX_TOL and rep[k]["mean_abs_255"] < IMG_MEAN_TOL, rep


def test_config5_hires_bislerp_10_steps_decode_1024(stack, conds):
    """hires-fix second pass (LD.py:10585-10603) from the reference's 64x64 result: LatentUpscale (bislerp on the device) ->
    KSampler2 10 Euler-a steps at denoise 0.45, cfg 8 on 128x128 latents -> 1024^2 decode."""
    from lightdiffusion_amd import nodes
    from lightdiffusion_amd import sampling as S
    model, _, vae = stack
    pos, neg, _ = conds
    g, g2 = load_golden("e2e_cfg5"), load_golden("e2e_cfg2")
    up = nodes.LatentUpscale(model.load_device).upscale({"samples": g2["latent"]}, "bislerp", 1024, 1024)[0]
    rep = {"bislerp_rel_l2": rel_l2(up["samples"], g["upscaled"])}
    tr = _Traj()
    noise = S.prepare_noise(up["samples"], 5005)
    ks = S.KSampler1(model, steps=10, device=model.load_device, sampler="euler_ancestral", scheduler="normal", denoise=0.45,
                     model_options=model.model_options)
    out = ks.sample(noise, pos, neg, cfg=float(g["cfg"]), latent_image=up["samples"], callback=tr, seed=5005).cpu()
    rep["latent_rel_l2"] = rel_l2(out, g["latent"])
    rep["traj_rel_l2"] = _curve(tr.xs, g["traj_sub"])
    # and through the node, which must agree bit for bit with the sampler-level call (same noise, same graph)
    out_node = nodes.KSampler2().sample(model, 5005, 10, float(g["cfg"]), "euler_ancestral", "normal", pos, neg, up, denoise=0.45)[0]["samples"]
    assert torch.equal(out_node, out)
    img = nodes.VAEDecode().decode(vae, {"samples": out})[0]
    assert img.shape == (1, 1024, 1024, 3)
    rep["image"] = _image_report(img, g)
    REPORT["cfg5"] = rep
    _dump()
    assert rep["bislerp_rel_l2"] < 1e-4
    assert rep["latent_rel_l2"] < LATENT_TOL["cfg5"] and max(rep["traj_rel_l2"]) < LATENT_TOL["cfg5"], rep
    assert rep["image"]["max_abs_255"] < IMG_MAX_TOL and rep["image"]["mean_abs_255"] < IMG_MEAN_TOL, rep


def test_config5_as_row0_of_batch4(stack, conds):
    """Config #5 at ITS batch (4 images => UNet batch 8 at 128x128 latents, the CFG-pair route at nb = 4 that bench.py times): the reference's
    B=1 run is row 0 of a batch of 4 (rows 1-3: other upscaled latents and other noise), driven through the sampler-level entry with an injected
    full-batch noise sampler (the pattern of the #3 test).  Bounds: the B=1 bound on the final latent, and — because the final latent is 92 % its own
    input — the same error relative to what the 10 steps CHANGED, ||latent - upscaled|| (0.084 ||latent|| in the golden)."""
    from lightdiffusion_amd import nodes
    from lightdiffusion_amd import sampling as S
    model, _, vae = stack
    pos, neg, _ = conds
    g, g2 = load_golden("e2e_cfg5"), load_golden("e2e_cfg2")
    up = nodes.LatentUpscale(model.load_device).upscale({"samples": g2["latent"]}, "bislerp", 1024, 1024)[0]["samples"].cpu()
    other = torch.Generator().manual_seed(77)
    lat4 = torch.cat([up, torch.roll(up, 17, -1), torch.flip(up, (-2,)), 0.9 * torch.roll(up, 31, -2)])
    noise4 = torch.cat([S.prepare_noise(up, 5005), torch.randn(3, 4, 128, 128, generator=other)])    # (prepare_noise seeds the global generator: the reference's order)

    def batch4_noise_sampler(sigma, sigma_next):
        ref_row = torch.randn(1, 4, 128, 128)                                 # the reference's draw: global generator, batch 1
        return torch.cat([ref_row, torch.randn(3, 4, 128, 128, generator=other)]).to(DEV)

    ks = S.KSampler1(model, steps=10, device=model.load_device, sampler="euler_ancestral", scheduler="normal", denoise=0.45,
                     model_options=model.model_options)
    tr = _Traj()
    out4 = S.sample(model, noise4, pos, neg, float(g["cfg"]), model.load_device, S.ksampler("euler_ancestral", {"noise_sampler": batch4_noise_sampler}),
                    ks.sigmas, model.model_options, latent_image=lat4, callback=tr, seed=5005).cpu()
    assert out4.shape == (4, 4, 128, 128) and torch.isfinite(out4).all()
    changed = float((g["latent"] - g["upscaled"]).norm())
    rep = {"row0_latent_rel_l2": rel_l2(out4[:1], g["latent"]), "row0_traj_rel_l2": _curve(tr.xs, g["traj_sub"]),
           "row0_err_over_change": float((out4[:1] - g["latent"]).norm()) / changed,
           "change_over_latent": changed / float(g["latent"].norm())}
    img = nodes.VAEDecode().decode(vae, {"samples": out4[:1]})[0]
    rep["image"] = _image_report(img, g)
    REPORT["cfg5_batch4"] = rep
    _dump()
    assert rep["row0_latent_rel_l2"] < LATENT_TOL["cfg5"] and max(rep["row0_traj_rel_l2"]) < LATENT_TOL["cfg5"], rep
    assert rep["row0_err_over_change"] < 1.2e-2, rep          # 1e-3 of the latent = 1.2 % of what the ten steps change
    assert rep["image"]["max_abs_255"] < IMG_MAX_TOL and rep["image"]["mean_abs_255"] < IMG_MEAN_TOL, rep

