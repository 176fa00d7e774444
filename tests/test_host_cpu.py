"""CPU: host-side logic of the product package (no HIP compute) against the reference-generated goldens, and the
N>1 path (broadcast / shard / gather) over gloo with world_size 2."""
import json
import os
import zlib

import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, load_golden, rel_l2
from lightdiffusion_amd import weights as W


def test_schedules_match_reference():
    from lightdiffusion_amd import sampling as S
    g = load_golden("schedules")
    ms = S.ModelSampling()
    assert torch.equal(ms.sigmas, g["sigmas"]) and torch.equal(ms.log_sigmas, g["log_sigmas"])
    assert torch.equal(S.calculate_sigmas(ms, "karras", 20), g["karras20"])
    assert torch.equal(S.calculate_sigmas(ms, "normal", 30), g["normal30"])
    ks = S.KSampler1.__new__(S.KSampler1)
    ks.model = type("M", (), {"get_model_object": staticmethod(lambda name: ms)})()
    ks.scheduler = "normal"
    ks.set_steps(10, 0.45)                                   # hires-fix: 22 normal sigmas, last 11 kept (LD.py:3097-3104)
    assert torch.equal(ks.sigmas, g["normal10_d045"])
    assert torch.equal(ms.timestep(g["probe_sigma"]), g["probe_t"]) and torch.equal(ms.sigma(g["tq"]), g["sigma_of_t"])
    for (a, b), (d, u) in zip(((14.6, 11.7), (1.0, 0.5), (0.05, 0.0)), g["anc"].tolist()):
        assert S.get_ancestral_step(a, b) == (d, u)
    with pytest.raises(ValueError):
        S.calculate_sigmas(ms, "exponential", 10)
    assert torch.equal(S.prepare_noise(torch.zeros(1, 4, 8, 8), 7), torch.randn(1, 4, 8, 8, generator=torch.manual_seed(7)))


def test_prompt_weights_and_chunking():
    from lightdiffusion_amd.clip import PromptTokenizer, escape_important, parse_prompt_weights
    g = json.load(open(os.path.join(GOLDEN, "prompt_weights.json")))
    for text, want in g.items():
        assert [list(t) for t in parse_prompt_weights(escape_important(text), 1.0)] == want
    word_ids = lambda w: [100 + zlib.crc32(f"{w}#{i}".encode()) % 40000 for i in range(1 + len(w) // 4)]
    tok = PromptTokenizer(word_ids)
    chunks = json.load(open(os.path.join(GOLDEN, "token_chunks.json")))
    for text, want in chunks.items():
        got = tok.tokenize_with_weights(text)
        assert [[list(p) for p in c] for c in got] == want, text
        assert all(len(c) == 77 and c[0][0] == 49406 for c in got)


class _OracleTextModel:
    """The text-model callable `clip.CLIP` drives, built from the oracle's CPU restatement (the package itself only has the HIP one)."""

    def __init__(self, cfg, sd):
        self.cfg, self.sd = cfg, sd

    def __call__(self, tokens, intermediate_output=None):
        from oracle import sd15_ref as O
        last = O.clip_text_model(self.sd, self.cfg, tokens, layer_idx=None)
        inter = None if intermediate_output is None else O.clip_text_model(self.sd, self.cfg, tokens, layer_idx=intermediate_output)
        pooled = last[torch.arange(last.shape[0]), tokens.to(torch.int).argmax(dim=-1)]
        return last, inter, pooled


def test_clip_text_model_and_weight_lerp():
    from lightdiffusion_amd.clip import CLIP
    g = load_golden("clip_tiny")
    cfg = W.tiny_clip_config()
    tm = _OracleTextModel(cfg, W.synth_state_dict(W.clip_param_shapes(cfg)))
    last, inter, pooled = tm(g["tokens"], intermediate_output=-2)
    assert rel_l2(last, g["last"]) < 5e-6 and rel_l2(inter, g["inter_m2"]) < 5e-6 and rel_l2(pooled, g["pooled"]) < 5e-6
    clip = CLIP(tm, None, layer_idx=-2)
    toks = g["tokens"][0].tolist()
    plain = clip.encode_from_tokens([[(t, 1.0) for t in toks]])
    assert rel_l2(plain, g["inter_m2"][:1]) < 5e-6
    cond, pooled1 = clip.encode_from_tokens([[(t, 1.3 if 2 <= i < 5 else 1.0) for i, t in enumerate(toks)]], return_pooled=True)
    empty = clip.encode_from_tokens([[(49406, 1.0)] + [(49407, 1.0)] * 76])
    assert torch.allclose(cond[0, 5:], plain[0, 5:], atol=1e-6)
    assert torch.allclose(cond[0, 2:5], (plain[0, 2:5] - empty[0, 2:5]) * 1.3 + empty[0, 2:5], atol=1e-5)
    assert cond.shape == (1, 77, cfg["hidden_size"]) and pooled1.shape == (1, cfg["hidden_size"])
    with pytest.raises(RuntimeError):
        clip.tokenize("needs a tokenizer")


def test_node_surface_shapes_and_errors():
    from lightdiffusion_amd import nodes, sampling
    lat = nodes.EmptyLatentImage().generate(512, 768, 3)[0]["samples"]
    assert lat.shape == (3, 4, 96, 64) and float(lat.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        sampling.ksampler("ddim")
    assert sampling.ksampler("dpm_adaptive").sampler_function is not None
    ctx = sampling._cat_ctx([torch.zeros(1, 77, 8), torch.ones(1, 154, 8)])      # lcm padding by repetition (LD.py:647-663)
    assert ctx.shape == (2, 154, 8)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import lightdiffusion_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        L.lib()


# ------------------------------------------------------------------ world_size 2 over gloo
def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from lightdiffusion_amd import dist as D
    r, w, _ = D.init("gloo")
    assert (r, w) == (rank, world)
    shapes = [(1, 77, 16), (1, 154, 16)]
    src = [torch.arange(77 * 16, dtype=torch.float32).view(shapes[0]), torch.full(shapes[1], 3.5)] if rank == 0 else [None, None]
    cond, uncond = D.broadcast_conditioning(src, shapes, src=0, device=torch.device("cpu"))
    ok = bool(cond[0, 1, 0] == 16.0) and bool((uncond == 3.5).all())
    rows = D.shard_rows(5, rank, world)
    noise = D.full_batch_noise((5, 4, 8, 8), 42, rows)
    full = torch.randn(5, 4, 8, 8, generator=torch.manual_seed(42))
    ok = ok and torch.equal(noise, full[rows])
    imgs = torch.full((rows.stop - rows.start, 2, 2, 3), float(rank))
    allimg = D.gather_images(imgs, dst=0)
    if rank == 0:
        ok = ok and allimg.shape[0] == 5 and allimg[:3].eq(0).all().item() and allimg[3:].eq(1).all().item()
    else:
        ok = ok and allimg is None
    q.put((rank, ok, (rows.start, rows.stop)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [(0, 3), (3, 5)]


# ------------------------------------------------------------------ the composed sharded entry (config #4) over gloo
class _StubCLIP:
    def tokenize(self, text):
        return text

    def encode_from_tokens(self, tokens, return_pooled=False):
        gen = torch.Generator().manual_seed(len(tokens) + 11)
        return torch.randn(1, 77 * (2 if len(tokens) > 10 else 1), 16, generator=gen)


class _StubModel:
    load_device = torch.device("cpu")
    model_options = {}

    def get_model_object(self, name):
        from lightdiffusion_amd import sampling as S
        return S.ModelSampling()


class _StubVAE:
    def decode(self, samples):
        return samples.permute(0, 2, 3, 1)[..., :3].contiguous()


def _stub_denoise(x, sigma, ctx):
    """row-wise toy denoiser that depends on the conditioning (so a wrong broadcast shows)"""
    return x * (1.0 / (1.0 + sigma.view(-1, 1, 1, 1) ** 2)) + 0.01 * ctx.mean()


def _run_sharded(global_batch):
    """txt2img_sharded with the device sampler loop replaced by the oracle's Euler-a (test seam `run_sampler`)."""
    from lightdiffusion_amd import nodes
    from lightdiffusion_amd import sampling as S
    from oracle import sd15_ref as O

    def run_sampler(noise, latent, pos, neg, sigmas, extra):
        ms = S.ModelSampling()
        x = ms.noise_scaling(sigmas[0], noise, latent, True)
        ctx = torch.cat([neg[0][0], pos[0][0][:, :77]])
        return O.sample_euler_ancestral(lambda xx, ss: _stub_denoise(xx, ss, ctx), x, [float(s) for s in sigmas],
                                        noise_sampler=extra["noise_sampler"])

    return nodes.txt2img_sharded(_StubModel(), _StubCLIP(), _StubVAE(), "a long prompt, two chunks", "", width=64, height=48,
                                 global_batch=global_batch, seed=321, steps=5, run_sampler=run_sampler)


def _sharded_worker(rank, world, port, q, global_batch=5):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from lightdiffusion_amd import dist as D
    D.init("gloo")
    torch.manual_seed(1000 + rank)           # ranks start from different global-generator states: the entry must not depend on them
    out = _run_sharded(global_batch)
    q.put((rank, None if out is None else out.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_txt2img_sharded_world_2_equals_single_process():
    """Config #4's call order (broadcast -> shard -> full-batch noise rows -> per-rank loop -> gather) on 2 gloo ranks reproduces
    the single-process Euler-a result row for row (uneven split 3 + 2; initial noise and every ancestral draw sliced)."""
    single = _run_sharded(5)
    assert single.shape == (5, 6, 8, 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 137) % 500
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res[1] is None and res[0].shape == single.shape
    assert torch.equal(res[0], single)


def _synthetic_checkpoint(tiny=True):
    ucfg, vcfg, ccfg = W.tiny_unet_config(), W.tiny_vae_config(), W.tiny_clip_config()
    sd = {}
    sd.update({"model.diffusion_model." + k: v.half() for k, v in W.synth_state_dict(W.unet_param_shapes(ucfg)).items()})
    vshapes = dict(W.vae_decoder_param_shapes(vcfg), **W.vae_encoder_param_shapes(vcfg))
    sd.update({"first_stage_model." + k: v for k, v in W.synth_state_dict(vshapes).items()})
    sd.update({"cond_stage_model.transformer." + k: v for k, v in W.synth_state_dict(W.clip_param_shapes(ccfg)).items()})
    return sd, ucfg, vcfg, ccfg


def test_checkpoint_config_detection_and_lora_merge():
    from lightdiffusion_amd import checkpoint as CK
    sd, ucfg, vcfg, ccfg = _synthetic_checkpoint()
    assert CK.detect_unet_config(sd) == ucfg
    full = {"model.diffusion_model." + k: torch.empty(s, device="meta") for k, s in W.unet_param_shapes(W.sd15_unet_config()).items()}
    assert CK.detect_unet_config(full) == W.sd15_unet_config()          # the real SD1.5 layout (LD.py:6065-6182 result)
    got, has_enc = CK.detect_vae_config(sd)
    assert got == vcfg and has_enc
    csd = CK.clip_state_dict(sd)
    assert CK.detect_clip_config(csd, num_heads=4) == ccfg
    key = "model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"
    before = sd[key].float().clone()
    up, down = torch.randn(64, 4), torch.randn(4, 64)
    name = "lora_unet_input_blocks_1_1_transformer_blocks_0_attn1_to_q"
    n = CK.merge_lora(sd, {name + ".lora_up.weight": up, name + ".lora_down.weight": down, name + ".alpha": torch.tensor(2.0)}, 0.5)
    assert n == 1 and n.unet == 1 and n.clip == 0 and torch.allclose(sd[key].float(), before + 0.5 * (2.0 / 4) * (up @ down), atol=2e-3)
    with pytest.raises(ValueError):
        CK.detect_unet_config({"foo": torch.zeros(1)})


def _lora_from_golden(g):
    return {k[len("lora::"):]: v for k, v in g.items() if k.startswith("lora::")}


def test_lora_key_maps_and_merge_match_reference():
    """f3: the diffusers -> ldm parameter map against the reference's `unet_to_diffusers` table (tiny and SD1.5 layouts), and a
    LoRA keyed the ways real files are keyed (kohya ldm / kohya diffusers / diffusers-native / conv / lora_te) merged into a
    checkpoint against weights patched by the reference's own ModelPatcher (oracle/make_golden.py `lora`)."""
    import warnings
    from lightdiffusion_amd import checkpoint as CK
    ref = json.load(open(os.path.join(GOLDEN, "unet_to_diffusers.json")))
    for tag, cfg in (("tiny", W.tiny_unet_config()), ("sd15", W.sd15_unet_config())):
        keys = [CK.UNET_PREFIX + k for k in W.unet_param_shapes(cfg)]
        have = {k[len(CK.UNET_PREFIX):] for k in keys}
        assert CK.unet_to_diffusers(keys, cfg["num_res_blocks"]) == {k: v for k, v in ref[tag].items() if v in have}
    g = load_golden("lora_tiny")
    lora = _lora_from_golden(g)
    sd, ucfg, vcfg, ccfg = _synthetic_checkpoint()
    sd.update({"model.diffusion_model." + k: v for k, v in W.synth_state_dict(W.unet_param_shapes(ucfg)).items()})   # fp32 base weights
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        res = CK.merge_lora(sd, lora, 0.8, 0.6)
    assert (res.unet, res.clip) == (len(g["patched_unet_keys"]), len(g["patched_clip_keys"])) == (8, 2)
    assert res.unmatched == ("lora_unet_not_a_layer_of_this_model",) and any("match no layer" in str(w.message) for w in wlist)
    P = CK.UNET_PREFIX
    assert torch.allclose(sd[P + "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"], g["w_attn1_to_q"], atol=1e-6)
    assert torch.allclose(sd[P + "input_blocks.1.0.in_layers.2.weight"], g["w_conv1"], atol=1e-6)          # 3x3 conv pair
    want = {"model." + str(k) for k in g["patched_unet_keys"]}
    km = CK.lora_key_map(sd)
    assert {km[m] for m in {k[:-len(".lora_up.weight")] for k in lora if k.endswith(".lora_up.weight")} if m in km and km[m].startswith(P)} == want


def test_txt2img_sharded_world_3_uneven_batch_8():
    """global batch 8 over 3 ranks (3 + 3 + 2): the padded all-gather path of `gather_images` and uneven row blocks."""
    single = _run_sharded(8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 271) % 500
    procs = [ctx.Process(target=_sharded_worker, args=(r, 3, port, q, 8)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res[1] is None and res[2] is None and torch.equal(res[0], single)
    from lightdiffusion_amd import dist as D
    assert [(D.shard_rows(8, r, 3).start, D.shard_rows(8, r, 3).stop) for r in range(3)] == [(0, 3), (3, 6), (6, 8)]
    assert [D.shard_rows(2, r, 3).stop - D.shard_rows(2, r, 3).start for r in range(3)] == [1, 1, 0]      # a rank may own nothing


def test_bench_spawn_ranks_environment(monkeypatch):
    """`bench.py --gpus N` without a torchrun environment: N children, each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    rendezvous on 127.0.0.1, dmabuf IPC kept; the parent never imports torch (it must not touch the GPU before its ranks do)."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("ld_bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    started = []

    class FakeProc:
        def __init__(self, argv, env):
            started.append((argv, env))

        def wait(self):
            return 0

    import types
    monkeypatch.setattr(bench, "subprocess", types.SimpleNamespace(Popen=lambda argv, env=None: FakeProc(argv, env)))
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(3, ["--gpus", "3", "--steps", "5"])
    assert e.value.code == 0 and len(started) == 3
    ports = set()
    for r, (argv, env) in enumerate(started):
        assert argv[0] == sys.executable and argv[1].endswith("bench.py") and argv[2:] == ["--gpus", "3", "--steps", "5"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "3", "127.0.0.1")
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1
    # the parent process of a --gpus N run imports no torch before it spawns (checked in a clean interpreter)
    code = ("import sys, runpy\nsys.argv = ['bench.py', '--gpus', '2']\nimport subprocess\n"
            "class P:\n    def __init__(self, *a, **k): assert 'torch' not in sys.modules, 'torch imported before spawn'\n"
            "    def wait(self): return 0\n"
            "subprocess.Popen = P\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules\nprint('parent-clean')\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=60)
    assert out.returncode == 0 and "parent-clean" in out.stdout, out.stderr


class _StubApply:
    def apply_model(self, *a, **k):
        raise AssertionError("the wrapper hook must be used")


def test_sampling_function_averages_multiple_conds():
    """calc_cond_batch with several entries per list (LD.py:2492-2591): one UNet call over (n_uncond + n_cond) * B samples in the
    reference's batch order, per-list average, then CFG; `area` / `strength` keys are inert (the reference's get_area_and_mult
    hard-codes the whole latent and weight 1, LD.py:2435-2458)."""
    from lightdiffusion_amd import sampling as S
    b, d = 2, 8
    calls = []

    def wrapper(apply_model, params):
        calls.append(params)
        ctx = params["c"]["c_crossattn"]
        return params["input"] * 0.5 + ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1)

    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, 4, 6, 5, generator=g)
    mk = lambda t, **kw: dict(kw, cross_attn=torch.randn(b, t, d, generator=g))
    pos = [mk(77), mk(77, area=(2, 2, 0, 0), strength=0.3)]
    neg = [mk(77), mk(154)]
    out = S.sampling_function(_StubApply(), x, torch.full((b,), 2.0), neg, pos, 6.0, {"model_function_wrapper": wrapper})
    assert len(calls) == 1 and calls[0]["cond_or_uncond"] == [1, 1, 0, 0] and calls[0]["input"].shape[0] == 4 * b
    ctx = calls[0]["c"]["c_crossattn"]
    assert ctx.shape == (4 * b, 154, d)
    assert torch.equal(ctx[:b], neg[1]["cross_attn"]) and torch.equal(ctx[b:2 * b], neg[0]["cross_attn"].repeat(1, 2, 1))
    assert torch.equal(ctx[2 * b:3 * b], pos[1]["cross_attn"].repeat(1, 2, 1)) and torch.equal(ctx[3 * b:], pos[0]["cross_attn"].repeat(1, 2, 1))
    m = lambda c: c["cross_attn"].mean(dim=(1, 2)).view(-1, 1, 1, 1)
    u = x * 0.5 + (m(neg[0]) + m(neg[1])) / 2
    c = x * 0.5 + (m(pos[0]) + m(pos[1])) / 2
    assert torch.allclose(out, u + (c - u) * 6.0, atol=1e-5)
    with pytest.raises(ValueError):
        S.sampling_function(_StubApply(), x, torch.full((b,), 2.0), [], pos, 6.0, {"model_function_wrapper": wrapper})
    with pytest.raises(RuntimeError):
        S.sampling_function(_StubApply(), x, torch.full((b,), 2.0), [dict(cross_attn=torch.zeros(3, 77, d))], pos, 6.0, {"model_function_wrapper": wrapper})


def test_lora_alias_patches_a_weight_once():
    """A file that names one weight through two aliases patches it once — the reference keys its patch dict by TARGET weight
    (LD.py:549-575); a module with lora_up but no lora_down is reported as such, not as 'matches no layer'."""
    import warnings
    from lightdiffusion_amd import checkpoint as CK
    sd, ucfg, _, _ = _synthetic_checkpoint()
    key = "model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"
    sd[key] = sd[key].float()
    before = sd[key].clone()
    g = torch.Generator().manual_seed(9)
    up, down = torch.randn(64, 4, generator=g), torch.randn(4, 64, generator=g)
    up2, down2 = torch.randn(64, 4, generator=g), torch.randn(4, 64, generator=g)
    ldm = "lora_unet_input_blocks_1_1_transformer_blocks_0_attn1_to_q"
    dif = "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q"
    km = CK.lora_key_map(sd)
    assert km[ldm] == km[dif] == key
    lora = {ldm + ".lora_up.weight": up, ldm + ".lora_down.weight": down, dif + ".lora_up.weight": up2, dif + ".lora_down.weight": down2,
            "lora_unet_input_blocks_1_1_transformer_blocks_0_attn1_to_k.lora_up.weight": up}
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        res = CK.merge_lora(sd, lora, 1.0)
    assert res == 1 and res.unet == 1 and res.unmatched == ()
    assert res.missing_down == ("lora_unet_input_blocks_1_1_transformer_blocks_0_attn1_to_k",)
    assert any("no lora_down" in str(w.message) for w in wl)
    winner = [n for n in km if km[n] == key and n in (ldm, dif)][-1]            # last alias in key-map order wins
    u, dn = (up, down) if winner == ldm else (up2, down2)
    assert torch.allclose(sd[key], before + u @ dn, atol=1e-5)


def test_bench_roofline_pricing():
    """bench.py prices a kernel against the roof its arithmetic intensity puts it under (ridge = 2.5 PFLOP/s / 8 TB/s), from the per-launch
    table of `ld_unet_profile_launches`, and reports PMC traffic only when the pass was taken with the library that is loaded."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ld_bench_roofs", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.launch_min_bytes("gemm", (128, 1280, 1280, 1)) == 2.0 * (128 * 1280 + 1280 * 1280 + 128 * 1280)
    assert bench.launch_min_bytes("conv3", (65536, 320, 2880, 1)) == 2.0 * (65536 * 320 + 320 * 2880 + 65536 * 320)     # raw pixels x Cin, not the im2col
    assert bench.launch_min_bytes("geglu", (4096, 10240, 1280, 1)) == 2.0 * (4096 * 1280 + 10240 * 1280 + 4096 * 5120)
    assert bench.launch_min_bytes("attention", (128, 4096, 4096, 40)) == 2.0 * 128 * (2 * 4096 * 40 + 2 * 4096 * 40)
    rows = [("conv3", (65536, 320, 2880, 1), 2.0 * 65536 * 320 * 2880, 130.0, "big_conv"),
            ("gemm", (128, 1280, 1280, 1), 2.0 * 128 * 1280 * 1280, 8.0, "skinny"), ("gemm", (128, 1280, 1280, 1), 2.0 * 128 * 1280 * 1280, 8.0, "skinny")]
    stale = lambda kern: (None, None, "profiles/pmc_traffic.json[x] lib abc (stale: loaded library is def)")
    dom, hbm = bench.rooflines(rows, {"big_conv": 0.130, "skinny": 0.016}, stale)
    assert dom["kernel"] == "big_conv" and dom["bound"] == "mfma" and dom["unit"] == "TFLOP/s"
    assert abs(dom["achieved"] - rows[0][2] / 130e-6 / 1e12) < 1e-6 and abs(dom["frac"] - dom["achieved"] / 2500.0) < 1e-9
    assert dom["traffic"] is None and "stale" in dom["traffic_from"]
    assert hbm["kernel"] == "skinny" and hbm["bound"] == "hbm" and hbm["unit"] == "GB/s" and hbm["launches"] == 2
    b = bench.launch_min_bytes("gemm", (128, 1280, 1280, 1))
    assert abs(hbm["achieved"] - 2 * b / 16e-6 / 1e9) < 1e-6 and hbm["intensity_flop_per_byte"] < bench.RIDGE


def test_real_clip_tokenizer_matches_reference():
    """PromptTokenizer.from_pretrained on the reference checkout's CLIP vocabulary (HuggingFace CLIPTokenizer underneath, as the reference uses)
    against SDTokenizer.tokenize_with_weights on the same files: BOS / EOS ids read from the tokenizer, emphasis weights, escapes, a prompt that
    spills into a second 77-token chunk.  The vocabulary does not travel with this repository: skipped where the checkout is absent."""
    tdir = os.path.join(os.path.dirname(os.environ.get("LD_REFERENCE", "/root/reference/LightDiffusion.py")), "_internal", "sd1_tokenizer")
    if not os.path.exists(os.path.join(tdir, "vocab.json")):
        pytest.skip("reference tokenizer files not present")
    pytest.importorskip("transformers")
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    from lightdiffusion_amd.clip import PromptTokenizer
    tok = PromptTokenizer.from_pretrained(tdir)
    want = json.load(open(os.path.join(GOLDEN, "token_chunks_real.json")))
    assert any(len(v) > 1 for v in want.values())
    for text, chunks in want.items():
        got = tok.tokenize_with_weights(text)
        assert [[list(p) for p in c] for c in got] == chunks, text
