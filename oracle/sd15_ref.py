"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement (plain torch CPU ops, fp32) of the reference's
SD1.5 denoise hot path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module; the product package never does.

Parity status: PINNED.  Every function below is checked against outputs of the reference's own classes
(executed in the build container by `oracle/extract_ref.py` + `oracle/make_golden.py`; fixtures in
`tests/golden/`), see tests/test_oracle_vs_golden.py.  The reference holds no tests or golden vectors
of its own.  One path stays unpinned: `dpmpp_2m_sde` with eta>0 and the default
BrownianTree noise (third-party `torchsde==0.2.6`, absent here) — pinned instead with eta=0 and with an
injected noise_sampler (supported argument, LD.py:1183-1192).

All citations are file:line into /root/reference/LightDiffusion.py (= LD.py).
State-dict keys are the SD1.x checkpoint names minus prefix (`input_blocks.1.0.in_layers.0.weight` ...).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ======================================================================== schedules

def make_sigmas_table(linear_start: float = 0.00085, linear_end: float = 0.012, n: int = 1000) -> torch.Tensor:
    """ModelSamplingDiscrete._register_schedule, LD.py:1300-1326 + make_beta_schedule 787-796.  Returns fp64."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    return ((1 - ac) / ac) ** 0.5


class ModelSampling:
    """EPS + ModelSamplingDiscrete, LD.py:1258-1351."""

    def __init__(self):
        s64 = make_sigmas_table()
        self.sigmas = s64.float()                 # set_sigmas LD.py:1324-1326: both rounded from the fp64 table
        self.log_sigmas = s64.log().float()
        self.sigma_data = 1.0

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma: torch.Tensor) -> torch.Tensor:          # LD.py:1336-1339
        d = sigma.log() - self.log_sigmas[:, None]
        return d.abs().argmin(dim=0).view(sigma.shape)

    def sigma(self, t: torch.Tensor) -> torch.Tensor:                 # LD.py:1341-1351
        t = torch.clamp(t.float(), min=0, max=len(self.sigmas) - 1)
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()


def sigmas_karras(n: int, sigma_min: float, sigma_max: float, rho: float = 7.0) -> torch.Tensor:
    """get_sigmas_karras, LD.py:831-837."""
    ramp = torch.linspace(0, 1, n)
    a, b = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    s = (b + ramp * (a - b)) ** rho
    return torch.cat([s, s.new_zeros([1])])


def sigmas_normal(ms: ModelSampling, steps: int) -> torch.Tensor:
    """normal_scheduler, LD.py:2639-2651."""
    ts = torch.linspace(ms.timestep(ms.sigma_max), ms.timestep(ms.sigma_min), steps)
    return torch.FloatTensor([float(ms.sigma(t)) for t in ts] + [0.0])


def calculate_sigmas(ms: ModelSampling, scheduler: str, steps: int, denoise: Optional[float] = None) -> torch.Tensor:
    """calculate_sigmas LD.py:3045-3054 + KSampler1.set_steps 3097-3104."""
    def calc(n):
        if scheduler == "karras":
            return sigmas_karras(n, float(ms.sigma_min), float(ms.sigma_max))
        if scheduler == "normal":
            return sigmas_normal(ms, n)
        raise ValueError(scheduler)
    if denoise is None or denoise > 0.9999:
        return calc(steps)
    return calc(int(steps / denoise))[-(steps + 1):]


# ======================================================================== UNet

def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """LD.py:803-812: cat(cos, sin), fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(x, sd, p, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(x, sd, p, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """attention_pytorch LD.py:3966-3978 (SDPA, no mask), written out explicitly."""
    b, lq, c = q.shape
    d = c // heads
    q, k, v = (t.view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (p @ v).transpose(1, 2).reshape(b, lq, c)


def resblock(x, emb, sd: SD, p: str):
    """ResBlock1._forward LD.py:5273-5287 (GN eps 1e-5)."""
    h = _conv(F.silu(_gn(x, sd, p + ".in_layers.0", 1e-5)), sd, p + ".in_layers.2")
    h = h + _lin(F.silu(emb), sd, p + ".emb_layers.1")[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, p + ".out_layers.0", 1e-5)), sd, p + ".out_layers.3")
    if (p + ".skip_connection.weight") in sd:
        x = _conv(x, sd, p + ".skip_connection", padding=0)
    return x + h


def cross_attention(x, ctx, sd: SD, p: str, heads: int):
    """CrossAttention.forward LD.py:4028-4036."""
    ctx = x if ctx is None else ctx
    o = attention(_lin(x, sd, p + ".to_q"), _lin(ctx, sd, p + ".to_k"), _lin(ctx, sd, p + ".to_v"), heads)
    return _lin(o, sd, p + ".to_out.0")


def transformer_block(x, ctx, sd: SD, p: str, heads: int):
    """BasicTransformerBlock._forward LD.py:4117-4162 (LN eps 1e-5) + GEGLU 4508-4515."""
    c = x.shape[-1]
    ln = lambda t, n: F.layer_norm(t, (c,), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)
    x = x + cross_attention(ln(x, "norm1"), None, sd, p + ".attn1", heads)
    x = x + cross_attention(ln(x, "norm2"), ctx, sd, p + ".attn2", heads)
    a, g = _lin(ln(x, "norm3"), sd, p + ".ff.net.0.proj").chunk(2, dim=-1)
    return _lin(a * F.gelu(g), sd, p + ".ff.net.2") + x


def spatial_transformer(x, ctx, sd: SD, p: str, heads: int):
    """SpatialTransformer.forward LD.py:4239-4262 (GN eps 1e-6, conv proj in/out)."""
    b, c, h, w = x.shape
    t = _conv(_gn(x, sd, p + ".norm", 1e-6), sd, p + ".proj_in", padding=0)
    t = t.permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = transformer_block(t, ctx, sd, p + ".transformer_blocks.0", heads)
    t = t.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return _conv(t, sd, p + ".proj_out", padding=0) + x


def unet_plan(cfg: dict) -> dict:
    """Static walk of UNetModel1.__init__ (LD.py:5379-5686): which layers each block holds."""
    mc, cm = cfg["model_channels"], cfg["channel_mult"]
    td_in, td_out = list(cfg["transformer_depth"]), list(cfg["transformer_depth_output"])
    inp: List[list] = [[("conv", "input_blocks.0.0")]]
    idx = 1
    for level, _ in enumerate(cm):
        for _ in range(cfg["num_res_blocks"][level]):
            layers = [("res", f"input_blocks.{idx}.0")]
            if td_in.pop(0) > 0:
                layers.append(("st", f"input_blocks.{idx}.1"))
            inp.append(layers)
            idx += 1
        if level != len(cm) - 1:
            inp.append([("down", f"input_blocks.{idx}.0.op")])
            idx += 1
    mid = [("res", "middle_block.0")]
    if cfg["transformer_depth_middle"] > 0:
        mid.append(("st", "middle_block.1"))
    mid.append(("res", "middle_block.2"))
    out: List[list] = []
    idx = 0
    for level, _ in list(enumerate(cm))[::-1]:
        for i in range(cfg["num_res_blocks"][level] + 1):
            layers = [("res", f"output_blocks.{idx}.0")]
            j = 1
            if td_out.pop() > 0:
                layers.append(("st", f"output_blocks.{idx}.{j}"))
                j += 1
            if level and i == cfg["num_res_blocks"][level]:
                layers.append(("up", f"output_blocks.{idx}.{j}.conv"))
            out.append(layers)
            idx += 1
    return dict(input=inp, middle=mid, output=out)


def unet_forward(sd: SD, cfg: dict, x: torch.Tensor, t: torch.Tensor, ctx: torch.Tensor) -> torch.Tensor:
    """UNetModel1.forward LD.py:5688-5767.  x [N,4,h,w], t [N] (timestep index as float), ctx [N,L,ctx_dim]."""
    heads = cfg["num_heads"]
    plan = unet_plan(cfg)
    emb = _lin(F.silu(_lin(timestep_embedding(t, cfg["model_channels"]).to(x.dtype), sd, "time_embed.0")), sd, "time_embed.2")

    def run(layers, h, out_hw=None):
        for kind, p in layers:
            if kind == "conv":
                h = _conv(h, sd, p)
            elif kind == "res":
                h = resblock(h, emb, sd, p)
            elif kind == "st":
                h = spatial_transformer(h, ctx, sd, p, heads)
            elif kind == "down":
                h = _conv(h, sd, p, stride=2)                                   # Downsample1 LD.py:5155-5186
            elif kind == "up":
                size = out_hw if out_hw is not None else (h.shape[2] * 2, h.shape[3] * 2)
                h = _conv(F.interpolate(h, size=size, mode="nearest"), sd, p)   # Upsample1 LD.py:5141-5152
        return h

    hs, h = [], x
    for layers in plan["input"]:
        h = run(layers, h)
        hs.append(h)
    h = run(plan["middle"], h)
    for layers in plan["output"]:
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(layers, h, tuple(hs[-1].shape[2:]) if hs else None)
    return _conv(F.silu(_gn(h, sd, "out.0", 1e-5)), sd, "out.2")


def apply_model(sd: SD, cfg: dict, ms: ModelSampling, x: torch.Tensor, sigma: torch.Tensor, ctx: torch.Tensor,
                half: bool = False) -> torch.Tensor:
    """BaseModel.apply_model LD.py:5828-5860 → denoised x0.  `half=True` mimics the reference dtype policy
    (fp16 weights/activations, LD.py:6418-6423, 5842-5859) with fp16 storage rounding at the boundary."""
    s = sigma.view(-1, 1, 1, 1)
    xc = x / (s ** 2 + ms.sigma_data ** 2) ** 0.5
    t = ms.timestep(sigma).float()
    if half:
        xc, ctx = xc.half().float(), ctx.half().float()
    eps = unet_forward(sd, cfg, xc, t, ctx)
    if half:
        eps = eps.half()
    return x - eps.float() * s


# ======================================================================== CFG + samplers

def sampling_function(denoise: Callable, x: torch.Tensor, sigma: torch.Tensor, cond: torch.Tensor,
                      uncond: torch.Tensor, cfg_scale: float) -> torch.Tensor:
    """sampling_function / calc_cond_batch / cfg_function LD.py:2492-2626.
    One batched call in order [uncond, cond] (list reversed at LD.py:2515), then uncond + (cond-uncond)*cfg."""
    b = x.shape[0]
    rep = lambda c: c if c.shape[0] == b else c.expand(b, -1, -1)
    out = denoise(torch.cat([x, x]), torch.cat([sigma, sigma]), torch.cat([rep(uncond), rep(cond)]))
    u, c = out.chunk(2)
    return u + (c - u) * cfg_scale


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """LD.py:844-850."""
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    return (sigma_to ** 2 - sigma_up ** 2) ** 0.5, sigma_up


def sample_euler_ancestral(model: Callable, x, sigmas, eta=1.0, s_noise=1.0, noise_sampler=None):
    """LD.py:907-941.  model(x, sigma[B]) -> denoised.  Default noise = torch.randn_like(x) from the global generator."""
    noise_sampler = (lambda s, sn: torch.randn_like(x)) if noise_sampler is None else noise_sampler
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * s_in)
        sd_, su = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        d = (x - den) / sigmas[i]
        x = x + d * (sd_ - sigmas[i])
        if sigmas[i + 1] > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * s_noise * su
    return x


def sample_dpmpp_2m_sde(model: Callable, x, sigmas, eta=1.0, s_noise=1.0, noise_sampler=None, solver_type="midpoint"):
    """LD.py:1174-1244.  eta=0 ⇒ deterministic DPM++ 2M.  eta>0 needs an injected noise_sampler (BrownianTree unpinned)."""
    if eta and noise_sampler is None:
        raise ValueError("eta>0 requires an explicit noise_sampler (torchsde BrownianTree is not available)")
    s_in = x.new_ones([x.shape[0]])
    old, h_last, h = None, None, None
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * s_in)
        if sigmas[i + 1] == 0:
            x = den
        else:
            t, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - t
            eh = eta * h
            x = sigmas[i + 1] / sigmas[i] * (-eh).exp() * x + (-h - eh).expm1().neg() * den
            if old is not None:
                r = h_last / h
                if solver_type == "heun":
                    x = x + ((-h - eh).expm1().neg() / (-h - eh) + 1) * (1 / r) * (den - old)
                else:
                    x = x + 0.5 * (-h - eh).expm1().neg() * (1 / r) * (den - old)
            if eta:
                x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * eh).expm1().neg().sqrt() * s_noise
        old, h_last = den, h
    return x


def sample_dpm_adaptive(model: Callable, x, sigma_min, sigma_max, order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0.0,
                        icoeff=1.0, dcoeff=0.0, accept_safety=0.81, eta=0.0):
    """sample_dpm_adaptive / DPMSolver.dpm_solver_adaptive / PIDStepSizeController, LD.py:944-1170 (no-noise variant: su = 0).
    Returns (x, info)."""
    sig = lambda t: t.neg().exp()
    ones = x.new_ones([x.shape[0]])
    eps_at = lambda xx, t: (xx - model(xx, sig(t) * ones)) / sig(t)
    t0, t1 = -torch.tensor(sigma_max).log(), -torch.tensor(sigma_min).log()
    fwd = bool(t1 > t0)
    h_step = abs(h_init) * (1 if fwd else -1)
    od = 1.5 if eta else order
    b1, b2, b3 = (pcoeff + icoeff + dcoeff) / od, -(pcoeff + 2 * dcoeff) / od, dcoeff / od
    errs: List[float] = []
    s, x_prev = t0, x
    info = {"steps": 0, "nfe": 0, "n_accept": 0, "n_reject": 0}
    while (s < t1 - 1e-5) if fwd else (s > t1 + 1e-5):
        t = torch.minimum(t1, s + h_step) if fwd else torch.maximum(t1, s + h_step)
        h = t - s
        e0 = eps_at(x, s)
        r1, r2 = 1 / 3, 2 / 3                      # both estimates use r1 = 1/3, so they share eps(s) and eps(s + h/3)
        sa, sb = s + r1 * h, s + r2 * h
        u1 = x - sig(sa) * (r1 * h).expm1() * e0
        e1 = eps_at(u1, sa)
        x_lo = x - sig(t) * h.expm1() * e0 - sig(t) / (2 * r1) * h.expm1() * (e1 - e0)                 # dpm_solver_2_step
        u2 = x - sig(sb) * (r2 * h).expm1() * e0 - sig(sb) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1) * (e1 - e0)
        e2 = eps_at(u2, sb)
        x_hi = x - sig(t) * h.expm1() * e0 - sig(t) / r2 * (h.expm1() / h - 1) * (e2 - e0)             # dpm_solver_3_step
        delta = torch.maximum(torch.tensor(atol), torch.tensor(rtol) * torch.maximum(x_lo.abs(), x_prev.abs()))
        err = float(torch.linalg.norm((x_lo - x_hi) / delta) / x.numel() ** 0.5)
        inv = 1 / (err + 1e-8)
        if not errs:
            errs = [inv, inv, inv]
        errs[0] = inv
        f = 1 + math.atan(errs[0] ** b1 * errs[1] ** b2 * errs[2] ** b3 - 1)
        ok = f >= accept_safety
        if ok:
            errs[2], errs[1] = errs[1], errs[0]
            x_prev, x, s = x_lo, x_hi, t
            info["n_accept"] += 1
        else:
            info["n_reject"] += 1
        h_step = h_step * f
        info["nfe"] += order
        info["steps"] += 1
    return x, info


LATENT_SCALE = 0.18215        # SD15.scale_factor LD.py:137-147


def ksample(denoise: Callable, ms: ModelSampling, seed: int, steps: int, cfg: float, sampler_name: str,
            scheduler: str, positive: torch.Tensor, negative: torch.Tensor, latent: torch.Tensor,
            denoise_strength: float = 1.0, sampler_opts: Optional[dict] = None) -> torch.Tensor:
    """common_ksampler LD.py:6657-6701 → sample1 → KSampler1 → CFGGuider.inner_sample 2926-2965 → KSAMPLER.sample 2738-2773."""
    g = torch.manual_seed(seed)                                             # prepare_noise LD.py:3145-3153
    noise = torch.randn(latent.size(), dtype=latent.dtype, generator=g, device="cpu")
    sigmas = calculate_sigmas(ms, scheduler, steps, denoise_strength)
    lat = latent
    if torch.count_nonzero(lat) > 0:                                        # LD.py:2938-2941
        lat = lat * LATENT_SCALE
    s0 = float(sigmas[0])
    max_denoise = math.isclose(float(ms.sigma_max), s0, rel_tol=1e-05) or s0 > float(ms.sigma_max)   # LD.py:2719-2722
    x = noise * (torch.sqrt(1.0 + sigmas[0] ** 2.0) if max_denoise else sigmas[0]) + lat              # EPS.noise_scaling 1267-1274
    model = lambda xx, ss: sampling_function(denoise, xx, ss, positive, negative, cfg)
    opts = sampler_opts or {}
    if sampler_name == "euler_ancestral":
        x = sample_euler_ancestral(model, x, sigmas, **opts)
    elif sampler_name == "dpmpp_2m_sde":
        x = sample_dpmpp_2m_sde(model, x, sigmas, **opts)
    elif sampler_name == "dpm_adaptive":                                     # dpm_adaptive_function, LD.py:2777-2797
        smin = sigmas[-1] if sigmas[-1] != 0 else sigmas[-2]
        x = sample_dpm_adaptive(model, x, float(smin), float(sigmas[0]), **opts)[0]
    else:
        raise ValueError(sampler_name)
    return x.float() / LATENT_SCALE                                         # process_latent_out LD.py:2965


# ======================================================================== VAE decoder

def vae_resblock(x, sd: SD, p: str):
    """ResnetBlock.forward LD.py:3560-3576 (GN eps 1e-6)."""
    h = _conv(F.silu(_gn(x, sd, p + ".norm1", 1e-6)), sd, p + ".conv1")
    h = _conv(F.silu(_gn(h, sd, p + ".norm2", 1e-6)), sd, p + ".conv2")
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".nin_shortcut", padding=0)
    return x + h


def vae_attn(x, sd: SD, p: str):
    """AttnBlock.forward LD.py:3630-3642 + pytorch_attention 3591-3602: one head, d = C."""
    b, c, h, w = x.shape
    n = _gn(x, sd, p + ".norm", 1e-6)
    q, k, v = (_conv(n, sd, f"{p}.{t}", padding=0).reshape(b, c, h * w).transpose(1, 2) for t in "qkv")
    o = attention(q, k, v, 1).transpose(1, 2).reshape(b, c, h, w)
    return x + _conv(o, sd, p + ".proj_out", padding=0)


def vae_decode(sd: SD, cfg: dict, z: torch.Tensor) -> torch.Tensor:
    """VAE.decode LD.py:6357-6381 ∘ AutoencodingEngine.decode 3470-3473 ∘ Decoder.forward 3857-3882.
    z [B,4,h,w] (already /0.18215) → [B,8h,8w,3] fp32 in [0,1]."""
    h = _conv(z, sd, "post_quant_conv", padding=0)
    h = _conv(h, sd, "decoder.conv_in")
    h = vae_resblock(h, sd, "decoder.mid.block_1")
    h = vae_attn(h, sd, "decoder.mid.attn_1")
    h = vae_resblock(h, sd, "decoder.mid.block_2")
    for lvl in reversed(range(len(cfg["ch_mult"]))):
        for b in range(cfg["num_res_blocks"] + 1):
            h = vae_resblock(h, sd, f"decoder.up.{lvl}.block.{b}")
        if lvl != 0:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"decoder.up.{lvl}.upsample.conv")
    h = _conv(F.silu(_gn(h, sd, "decoder.norm_out", 1e-6)), sd, "decoder.conv_out")
    return torch.clamp((h + 1.0) / 2.0, 0.0, 1.0).movedim(1, -1)


def vae_encode_moments(sd: SD, cfg: dict, pixels: torch.Tensor) -> torch.Tensor:
    """VAE.encode up to the regularizer (LD.py:6383-6410 ∘ AutoencodingEngine.encode 3475-3481 ∘ Encoder.forward 3731-3758):
    pixels [B,H,W,3] in [0,1] -> moments [B,2z,H/8,W/8] (mean | logvar)."""
    h = _conv(pixels[..., :3].movedim(-1, 1) * 2.0 - 1.0, sd, "encoder.conv_in")
    nl = len(cfg["ch_mult"])
    for lvl in range(nl):
        for b in range(cfg["num_res_blocks"]):
            h = vae_resblock(h, sd, f"encoder.down.{lvl}.block.{b}")
        if lvl != nl - 1:      # Downsample, LD.py:3514-3528: zero-pad right/bottom by one, 3x3 stride 2, no padding
            h = _conv(F.pad(h, (0, 1, 0, 1)), sd, f"encoder.down.{lvl}.downsample.conv", stride=2, padding=0)
    h = vae_resblock(h, sd, "encoder.mid.block_1")
    h = vae_attn(h, sd, "encoder.mid.attn_1")
    h = vae_resblock(h, sd, "encoder.mid.block_2")
    h = _conv(F.silu(_gn(h, sd, "encoder.norm_out", 1e-6)), sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def vae_sample(moments: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample (LD.py:166-179): mean + std * randn from the host global generator."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * torch.randn(mean.shape)


# ======================================================================== CLIP-L

def clip_text_model(sd: SD, cfg: dict, tokens: torch.Tensor, layer_idx: Optional[int] = -2) -> torch.Tensor:
    """CLIPTextModel_ LD.py:4413-4463 (+CLIPLayer 4322-4349): causal, quick_gelu, LN 1e-5; returns the hidden
    state after layer `layer_idx` passed through final_layer_norm (clip-skip, LD.py:6604-6608), or the last one."""
    hdim, heads = cfg["hidden_size"], cfg["num_attention_heads"]
    P = "text_model."
    x = sd[P + "embeddings.token_embedding.weight"][tokens] + sd[P + "embeddings.position_embedding.weight"]
    L = x.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    ln = lambda t, p: F.layer_norm(t, (hdim,), sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    nl = cfg["num_hidden_layers"]
    stop = None if layer_idx is None else (nl + layer_idx if layer_idx < 0 else layer_idx)
    inter = None
    for i in range(nl):
        p = f"{P}encoder.layers.{i}"
        n = ln(x, p + ".layer_norm1")
        q, k, v = (_lin(n, sd, f"{p}.self_attn.{t}_proj") for t in "qkv")
        b = q.shape[0]
        d = hdim // heads
        q, k, v = (t.view(b, L, heads, d).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d) + mask, dim=-1) @ v
        x = x + _lin(a.transpose(1, 2).reshape(b, L, hdim), sd, p + ".self_attn.out_proj")
        m = _lin(ln(x, p + ".layer_norm2"), sd, p + ".mlp.fc1")
        x = x + _lin(m * torch.sigmoid(1.702 * m), sd, p + ".mlp.fc2")
        if i == stop:
            inter = x.clone()
    x = ln(x, P + "final_layer_norm")
    return x if inter is None else ln(inter, P + "final_layer_norm")


def encode_token_weights(encode: Callable, token_weight_pairs: Sequence[Sequence], empty_tokens: Sequence[int]) -> torch.Tensor:
    """ClipTokenWeightEncoder.encode_token_weights LD.py:4540-4569: per-token lerp against the empty prompt."""
    to_encode = [[t for t, _ in sec] for sec in token_weight_pairs]
    has_w = any(w != 1.0 for sec in token_weight_pairs for _, w in sec)
    if has_w or not to_encode:
        to_encode = to_encode + [list(empty_tokens)]
    out = encode(torch.tensor(to_encode, dtype=torch.long))
    outs = []
    for k in range(len(token_weight_pairs)):
        z = out[k:k + 1].clone()
        if has_w:
            w = torch.tensor([w for _, w in token_weight_pairs[k]], dtype=z.dtype)[None, :, None]
            z = torch.where(w != 1.0, (z - out[-1][None]) * w + out[-1][None], z)
        outs.append(z)
    return torch.cat(outs, dim=-2)


# ======================================================================== hires pre-step

def _slerp_rows(a: torch.Tensor, b: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """Row-wise spherical blend used by the reference's latent upscale (LD.py:430-463).
    a, b [R,C]; r [R,1].  Directions are slerped, magnitudes lerped; (anti)parallel rows fall back to a / lerp."""
    na, nb = a.norm(dim=1, keepdim=True), b.norm(dim=1, keepdim=True)
    ua = torch.where(na > 0, a / na, torch.zeros_like(a))
    ub = torch.where(nb > 0, b / nb, torch.zeros_like(b))
    cosw = (ua * ub).sum(dim=1, keepdim=True)
    w = torch.acos(cosw)
    sw = torch.sin(w)
    out = (torch.sin((1.0 - r) * w) / sw) * ua + (torch.sin(r * w) / sw) * ub
    out = out * (na * (1.0 - r) + nb * r)
    out = torch.where(cosw > 1 - 1e-5, a, out)
    return torch.where(cosw < 1e-5 - 1, a * (1.0 - r) + b * r, out)


def _bilinear_taps(n_src: int, n_dst: int):
    """Left tap, right tap and blend ratio of a half-pixel-centre (align_corners=False) bilinear resize —
    what LD.py:465-486 obtains by resizing an index ramp."""
    pos = ((torch.arange(n_dst, dtype=torch.float32) + 0.5) * (n_src / n_dst) - 0.5).clamp_(min=0.0)
    lo = pos.floor().clamp_(max=n_src - 1)
    frac = torch.where(lo >= n_src - 1, torch.zeros_like(pos), pos - lo)
    lo = lo.long()
    return lo, (lo + 1).clamp_(max=n_src - 1), frac


def bislerp(samples: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """bislerp LD.py:429-518 (the only `common_upscale` mode, LD.py:521-523): separable 2-tap resize, first along W
    then along H, blending the C-vector of each tap pair with `_slerp_rows`."""
    x = samples.float().permute(0, 2, 3, 1)                       # N,H,W,C
    n, h, w, c = x.shape
    lo, hi, fr = _bilinear_taps(w, width)
    y = _slerp_rows(x[:, :, lo].reshape(-1, c), x[:, :, hi].reshape(-1, c),
                    fr.view(1, 1, -1, 1).expand(n, h, -1, 1).reshape(-1, 1)).view(n, h, width, c)
    lo, hi, fr = _bilinear_taps(h, height)
    z = _slerp_rows(y[:, lo].reshape(-1, c), y[:, hi].reshape(-1, c),
                    fr.view(1, -1, 1, 1).expand(n, -1, width, 1).reshape(-1, 1)).view(n, height, width, c)
    return z.permute(0, 3, 1, 2).to(samples.dtype)
