"""Single-file SD1.x checkpoint ingestion: key layout and architecture detection.

Key layout the reference loads (LD.py:5921-5922, 5980-6009, 6446-6465):
  model.diffusion_model.*                      UNet
  first_stage_model.{encoder,decoder,quant_conv,post_quant_conv}.*   VAE
  cond_stage_model.transformer.[text_model.]*  CLIP-L
The architecture is inferred from which keys exist and their shapes, like the reference's detect_unet_config
(LD.py:6065-6182) — restated here for the SD1.x family only (no video / SDXL branches).  LoRA: the reference merges
`W += alpha * up @ down` into the weights before sampling (LD.py:3335-3354, 3407-3424); `merge_lora` does that merge on the
state dict, so the HIP side only ever sees final weights.
"""
from __future__ import annotations

import re
from typing import Dict, Optional, Tuple

import torch

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."
CLIP_PREFIX = "cond_stage_model.transformer."


def _count(keys, pattern: str) -> int:
    """number of distinct integer indices i for which a key matches pattern.format(i) as a prefix"""
    rx = re.compile("^" + re.escape(pattern).replace(r"\{\}", r"(\d+)"))
    idx = {int(m.group(1)) for k in keys for m in [rx.match(k)] if m}
    return max(idx) + 1 if idx else 0


def detect_unet_config(sd: Dict[str, torch.Tensor], prefix: str = UNET_PREFIX, num_heads: int = 8) -> dict:
    keys = [k[len(prefix):] for k in sd if k.startswith(prefix)]
    ks = set(keys)
    if "input_blocks.0.0.weight" not in ks:
        raise ValueError("not an SD1.x checkpoint: no UNet keys under '%s'" % prefix)
    get = lambda k: sd[prefix + k]
    mc, cin = get("input_blocks.0.0.weight").shape[:2]
    depth_of = lambda p: _count(keys, p + ".transformer_blocks.{}.")
    channel_mult, num_res, td_in = [], [], []
    level_res, level_mult, context_dim = 0, 0, None
    for i in range(1, _count(keys, "input_blocks.{}.")):
        if f"input_blocks.{i}.0.op.weight" in ks:
            channel_mult.append(level_mult)
            num_res.append(level_res)
            level_res = 0
        else:
            level_res += 1
            level_mult = get(f"input_blocks.{i}.0.out_layers.3.weight").shape[0] // mc
            d = depth_of(f"input_blocks.{i}.1")
            td_in.append(d)
            if d and context_dim is None:
                context_dim = get(f"input_blocks.{i}.1.transformer_blocks.0.attn2.to_k.weight").shape[1]
    channel_mult.append(level_mult)
    num_res.append(level_res)
    n_out = _count(keys, "output_blocks.{}.")
    # the reference builds this list back to front and pops from its end while constructing (LD.py:6107-6150, 5621)
    td_out = [depth_of(f"output_blocks.{j}.1") for j in reversed(range(n_out))]
    if max(td_in + td_out + [0]) > 1:
        raise ValueError("transformer depth > 1 is not an SD1.x UNet")
    return dict(in_channels=int(cin), out_channels=int(get("out.2.weight").shape[0]), model_channels=int(mc), channel_mult=channel_mult,
                num_res_blocks=num_res, transformer_depth=td_in, transformer_depth_output=td_out,
                transformer_depth_middle=depth_of("middle_block.1"), context_dim=int(context_dim), num_heads=num_heads)


def detect_vae_config(sd: Dict[str, torch.Tensor], prefix: str = VAE_PREFIX) -> Tuple[dict, bool]:
    keys = [k[len(prefix):] for k in sd if k.startswith(prefix)]
    get = lambda k: sd[prefix + k]
    ch = int(get("decoder.norm_out.weight").shape[0])
    levels = _count(keys, "decoder.up.{}.")
    mult = [int(get(f"decoder.up.{l}.block.0.conv1.weight").shape[0]) // ch for l in range(levels)]
    cfg = dict(z_channels=int(get("post_quant_conv.weight").shape[0]), ch=ch, ch_mult=mult,
               num_res_blocks=_count(keys, "decoder.up.0.block.{}.") - 1, out_ch=int(get("decoder.conv_out.weight").shape[0]))
    return cfg, any(k.startswith("encoder.") for k in keys)


def clip_state_dict(sd: Dict[str, torch.Tensor], prefix: str = CLIP_PREFIX) -> Dict[str, torch.Tensor]:
    """`cond_stage_model.transformer.*` -> keys under `text_model.` (sm_SD15.process_clip_state_dict, LD.py:5980-6009)."""
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            k2 = k[len(prefix):]
            k2 = k2 if k2.startswith("text_model.") else "text_model." + k2
            if not k2.endswith("position_ids"):
                out[k2] = v
    return out


def detect_clip_config(csd: Dict[str, torch.Tensor], num_heads: int = 12) -> dict:
    h = int(csd["text_model.embeddings.token_embedding.weight"].shape[1])
    return dict(vocab_size=int(csd["text_model.embeddings.token_embedding.weight"].shape[0]), hidden_size=h,
                intermediate_size=int(csd["text_model.encoder.layers.0.mlp.fc1.weight"].shape[0]),
                num_hidden_layers=_count(csd.keys(), "text_model.encoder.layers.{}."), num_attention_heads=num_heads,
                max_position_embeddings=int(csd["text_model.embeddings.position_embedding.weight"].shape[0]), hidden_act="quick_gelu")


# ------------------------------------------------------------------ LoRA ingestion
_RES_RENAME = {"in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj", "out_layers.0": "norm2",
               "out_layers.3": "conv2", "skip_connection": "conv_shortcut"}
_TOP_RENAME = {"input_blocks.0.0": "conv_in", "out.0": "conv_norm_out", "out.2": "conv_out", "time_embed.0": "time_embedding.linear_1",
               "time_embed.2": "time_embedding.linear_2"}


def _diffusers_module(ldm: str, num_res_blocks) -> Optional[str]:
    """ldm module path (no .weight/.bias) -> diffusers module path, for the SD1.x UNet layout.  The reference builds the
    forward table (`unet_to_diffusers`, LD.py:302-394); this is the inverse rule, read off the block numbering: input block
    1 + sum_{y<x}(R_y + 1) + i is resnet / attention i of down block x, the block after a level's last resnet is its
    downsampler; output blocks count the same way over the reversed levels, and the last block of a level carries the upsampler."""
    for k, v in _TOP_RENAME.items():
        if ldm == k:
            return v
    m = re.match(r"^(input_blocks|output_blocks|middle_block)\.(\d+)\.(.*)$", ldm)
    if not m:
        return None
    kind, n, rest = m.group(1), int(m.group(2)), m.group(3)

    def res_or_attn(prefix, i, rest):
        slot, _, sub = rest.partition(".")
        if slot == "0":
            for k, v in _RES_RENAME.items():
                if sub == k:
                    return f"{prefix}.resnets.{i}.{v}"
            return None
        return f"{prefix}.attentions.{i}.{sub}" if sub and not sub.startswith("conv") else None

    if kind == "middle_block":
        if n == 1:
            return f"mid_block.attentions.0.{rest}"
        for k, v in _RES_RENAME.items():
            if rest == k:
                return f"mid_block.resnets.{n // 2}.{v}"
        return None
    if kind == "input_blocks":
        base = 1
        for x, r in enumerate(num_res_blocks):
            if n < base + r:
                return res_or_attn(f"down_blocks.{x}", n - base, rest)
            if n == base + r:
                return f"down_blocks.{x}.downsamplers.0.conv" if rest == "0.op" else None
            base += r + 1
        return None
    base = 0
    for x, r in enumerate(reversed(list(num_res_blocks))):
        if n < base + r + 1:
            i = n - base
            if rest.endswith(".conv") and rest.split(".")[0] in ("1", "2"):
                return f"up_blocks.{x}.upsamplers.0.conv" if i == r else None
            return res_or_attn(f"up_blocks.{x}", i, rest)
        base += r + 1
    return None


def unet_to_diffusers(sd_keys, num_res_blocks, prefix: str = UNET_PREFIX) -> Dict[str, str]:
    """{diffusers parameter name: ldm parameter name} for every UNet parameter present in `sd_keys` (prefix stripped)."""
    out = {}
    for k in sd_keys:
        if not k.startswith(prefix):
            continue
        k = k[len(prefix):]
        mod, _, leaf = k.rpartition(".")
        d = _diffusers_module(mod, num_res_blocks)
        if d is not None:
            out[f"{d}.{leaf}"] = k
    return out


_CLIP_LORA_LAYERS = ("mlp.fc1", "mlp.fc2", "self_attn.k_proj", "self_attn.q_proj", "self_attn.v_proj", "self_attn.out_proj")


def lora_key_map(sd: Dict[str, torch.Tensor], unet_prefix: str = UNET_PREFIX, clip_prefix: str = CLIP_PREFIX) -> Dict[str, str]:
    """{name a LoRA file may use: checkpoint key it patches} — the naming schemes the reference accepts
    (model_lora_keys_unet / model_lora_keys_clip, LD.py:577-629): kohya names flattened from the ldm key or from the
    diffusers key (`lora_unet_...`), diffusers-native attention-processor names (with and without `unet.`), and for the text
    encoder `lora_te_...`, `lora_te1_...`, `text_encoder....`."""
    km = {}
    unet_keys = [k for k in sd if k.startswith(unet_prefix) and k.endswith(".weight")]
    for k in unet_keys:
        km["lora_unet_" + k[len(unet_prefix):-len(".weight")].replace(".", "_")] = k
    nrb = None
    try:
        nrb = detect_unet_config(sd, unet_prefix)["num_res_blocks"]
    except (ValueError, KeyError):
        pass
    if nrb is not None:
        for dk, lk in unet_to_diffusers(unet_keys, nrb, unet_prefix).items():
            mod = dk[:-len(".weight")]
            km["lora_unet_" + mod.replace(".", "_")] = unet_prefix + lk
            native = mod.replace(".to_", ".processor.to_")
            if native.endswith(".to_out.0"):
                native = native[:-2]
            km[native] = km["unet." + native] = unet_prefix + lk
    rx = re.compile("^" + re.escape(clip_prefix) + r"(?:text_model\.)?encoder\.layers\.(\d+)\.(.+)\.weight$")
    for k in sd:
        m = rx.match(k)
        if m and m.group(2) in _CLIP_LORA_LAYERS:
            b, c = m.group(1), m.group(2)
            for fmt in ("lora_te_text_model_encoder_layers_{}_{}", "lora_te1_text_model_encoder_layers_{}_{}"):
                km[fmt.format(b, c.replace(".", "_"))] = k
            km[f"text_encoder.text_model.encoder.layers.{b}.{c}"] = k
    return km


class LoraMergeResult(int):
    """number of merged tensors (an int, as before) with the breakdown attached"""
    unet = 0
    clip = 0
    unmatched: tuple = ()
    missing_down: tuple = ()


def merge_lora(sd: Dict[str, torch.Tensor], lora: Dict[str, torch.Tensor], strength: float = 1.0, strength_clip: Optional[float] = None,
               prefix: str = UNET_PREFIX, clip_prefix: str = CLIP_PREFIX) -> LoraMergeResult:
    """Merge a LoRA file into the checkpoint state dict in place, UNet and text encoder: for every pair the key map resolves,
    W += strength * (alpha / rank) * up @ down, in fp32, cast back to W's dtype (load_lora LD.py:549-575 + calculate_weight
    LD.py:3407-3424; the reference patches before sampling, LD.py:3335-3354, so the HIP side only ever sees final weights).
    `strength_clip` defaults to `strength` (load_lora_for_models takes both, LD.py:6203-6219).  LoRA modules that match
    nothing are reported (`.unmatched`) and warned about instead of being dropped silently."""
    import warnings
    km = lora_key_map(sd, prefix, clip_prefix)
    sc = strength if strength_clip is None else strength_clip
    n_unet = n_clip = 0
    seen = set()
    modules = sorted({k[:-len(".lora_up.weight")] for k in lora if k.endswith(".lora_up.weight")})
    present = set(modules)
    no_down = tuple(m for m in modules if m + ".lora_down.weight" not in lora)
    # one patch per TARGET weight, as the reference keys its patch_dict (`patch_dict[to_load[x]] = ...`, LD.py:549-575, iterating the
    # key map): a file that names one weight through two aliases (ldm- and diffusers-flattened `lora_unet_*`, `unet.` + bare
    # processor key) patches it once — the alias that comes last in key-map order wins
    patches: Dict[str, str] = {}
    for name, key in km.items():
        if name in present and name + ".lora_down.weight" in lora:
            patches[key] = name
            seen.add(name)
    for key, name in patches.items():
        up, down = lora[name + ".lora_up.weight"], lora[name + ".lora_down.weight"]
        is_clip = key.startswith(clip_prefix)
        scale = sc if is_clip else strength
        if name + ".alpha" in lora:
            scale = scale * float(lora[name + ".alpha"]) / down.shape[0]
        w = sd[key]
        delta = torch.mm(up.flatten(start_dim=1).float(), down.flatten(start_dim=1).float()).reshape(w.shape)
        sd[key] = (w.float() + scale * delta).to(w.dtype)
        n_clip += is_clip
        n_unet += not is_clip
    res = LoraMergeResult(n_unet + n_clip)
    res.unet, res.clip = n_unet, n_clip
    res.unmatched = tuple(m for m in modules if m not in seen and m not in no_down)
    res.missing_down = no_down
    if no_down:
        warnings.warn(f"LoRA: {len(no_down)} module(s) have lora_up but no lora_down weight and were not applied: " + ", ".join(no_down[:4]))
    if res.unmatched:
        warnings.warn(f"LoRA: {len(res.unmatched)} module(s) match no layer of this checkpoint and were not applied: "
                      + ", ".join(res.unmatched[:4]) + (" ..." if len(res.unmatched) > 4 else ""))
    return res


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    return ck.get("state_dict", ck)
