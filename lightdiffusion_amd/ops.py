"""Operator-level host mirror of the reference's `operations=` namespace (LD.py:2342-2429) and
`optimized_attention` (LD.py:3966-3988), backed by the HIP library through the C ABI.

Tensors are torch CUDA tensors used as *device memory only*: fp16, channels-last activations
([N,H,W,C] == [N, H*W, C] tokens), weights as the checkpoint stores them.  Every function raises
`LDError` on a non-zero status; none falls back to PyTorch math.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from ._lib import F16, F32, check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return t.data_ptr()


_WS = {}


def _ws(nbytes: int, device) -> torch.Tensor:
    """Scratch for split-K partials / GEGLU repack / LN-fold temporaries, cached per (device, stream) and grown on demand: the
    operator seam is also the CLIP path (60+ calls per prompt), so no per-call allocation.  Reuse is stream-ordered — every op
    that takes the buffer runs on the stream the buffer is keyed by and is done with it when the next one starts — so ops
    issued on two streams never share slabs.  Under hipGraph capture nothing is cached: a buffer first grown there would live in
    the graph's private pool (and a cached pointer baked into a graph would race with eager ops), so capture allocates per call."""
    d = torch.device(device)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           act: str = "none", alpha: float = 1.0) -> torch.Tensor:
    """y = act(alpha * x @ weight.T + bias) + residual.  x [..., K] fp16, weight [N, K] fp16 (nn.Linear layout).
    act='geglu': weight/bias rows are [value | gate] as in the checkpoint (LD.py:4508-4515); y has N/2 columns."""
    code = {"none": 0, "silu": 1, "geglu": 2, "quick_gelu": 3}[act]
    K = x.shape[-1]
    N = weight.shape[0]
    M = x.numel() // K
    y = torch.empty(*x.shape[:-1], N // 2 if code == 2 else N, dtype=torch.float16, device=x.device)
    ws = _ws(64 << 20, x.device)
    check(lib().ld_op_linear(_p(x), _p(weight), _p(bias), _p(residual), _p(y), M, N, K, alpha, code, _p(ws), ws.numel(), _stream()),
          "ld_op_linear")
    return y


def linear_ln(x: torch.Tensor, w_prod: torch.Tensor, b_prod: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
              weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float = 1e-5):
    """The UNet's LayerNorm fold as an operator pair: t = x @ w_prod.T + b_prod;  y = LayerNorm(t) @ weight.T + bias with the
    normalisation finished on the GEMM accumulators (no LayerNorm launch, no normalised tensor).  Returns (t, y)."""
    M, C = x.shape
    N = weight.shape[0]
    t = torch.empty(M, C, dtype=torch.float16, device=x.device)
    y = torch.empty(M, N, dtype=torch.float16, device=x.device)
    ws = _ws(2 * N * C + 8 * N + 8 * ((C + 63) // 64) * M + 4096, x.device)
    check(lib().ld_op_linear_ln(_p(x), _p(w_prod), _p(b_prod), _p(gamma), _p(beta), _p(weight), _p(bias), _p(t), _p(y), M, C, N, eps,
                                _p(ws), ws.numel(), _stream()), "ld_op_linear_ln")
    return t, y


def linear_ln_geglu(x: torch.Tensor, w_prod: torch.Tensor, b_prod: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                    weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5):
    """linear_ln with the GEGLU of FeedForward.net[0] as the consumer: t = x @ w_prod.T + b_prod;  [a | g] = LayerNorm(t) @ weight.T + bias;
    y = a * gelu(g) — the transformer block's MLP input as the executor runs it.  Returns (t, y) with y of width N / 2."""
    M, C = x.shape
    N = weight.shape[0]
    t = torch.empty(M, C, dtype=torch.float16, device=x.device)
    y = torch.empty(M, N // 2, dtype=torch.float16, device=x.device)
    ws = _ws(4 * N * C + 12 * N + 8 * ((C + 63) // 64) * M + 4096, x.device)
    check(lib().ld_op_linear_ln_geglu(_p(x), _p(w_prod), _p(b_prod), _p(gamma), _p(beta), _p(weight), _p(bias), _p(t), _p(y), M, C, N, eps,
                                      _p(ws), ws.numel(), _stream()), "ld_op_linear_ln_geglu")
    return t, y


def repack_conv_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """[O,I,kh,kw] (fp16/fp32) -> fp16 [O, kh*kw*I] tap-major / channel-minor, the layout the conv kernels read."""
    O, I, kh, kw = w_oihw.shape
    w_oihw = w_oihw.contiguous()
    out = torch.empty(O, kh * kw * I, dtype=torch.float16, device=w_oihw.device)
    if kh == 3:
        check(lib().ld_op_repack_conv(_p(w_oihw), F32 if w_oihw.dtype == torch.float32 else F16, O, I, _p(out), _stream()),
              "ld_op_repack_conv")
    else:
        out.copy_(w_oihw.reshape(O, I))
    return out


def conv2d(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], ksize: int = 3, stride: int = 1,
           x2: Optional[torch.Tensor] = None, out_hw: Optional[tuple] = None, rowvec: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NHWC conv (pad ksize//2) over the channel concat of x [N,H,W,C1] and optional x2 [N,H,W,C2];
    `out_hw` first resizes the input nearest-neighbour (Upsample1, LD.py:5141-5152); rowvec [N,Cout] is added per image."""
    n, h, w, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    hv, wv = (h, w) if out_hw is None else out_hw
    cout = w_packed.shape[0]
    ho = (hv - 1) // stride + 1 if ksize == 3 else hv
    wo = (wv - 1) // stride + 1 if ksize == 3 else wv
    y = torch.empty(n, ho, wo, cout, dtype=torch.float16, device=x.device)
    ws = _ws(192 << 20, x.device)
    check(lib().ld_op_conv(_p(x), c1, _p(x2), c2, n, h, w, hv, wv, stride, ksize, _p(w_packed), _p(bias), _p(rowvec), _p(residual),
                           _p(y), cout, _p(ws), ws.numel(), _stream()), "ld_op_conv")
    return y


def conv2d_skip(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, s1: torch.Tensor, s2: Optional[torch.Tensor], w_skip: torch.Tensor,
                b_skip: torch.Tensor, rowvec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ResBlock1's out_layers convolution + its 1x1 skip_connection as one contraction (LD.py:5267, 5273-5287):
    y = conv3x3(x; w_packed) + bias + conv1x1(cat(s1, s2); w_skip) + b_skip.  x [N,H,W,C]; s1 / s2 raw NHWC sources of the same H x W;
    w_skip [Cout, C(s1) + C(s2)] (the checkpoint's [O, I, 1, 1] reshaped)."""
    n, h, w, c = x.shape
    sc1 = s1.shape[-1]
    sc2 = 0 if s2 is None else s2.shape[-1]
    cout = w_packed.shape[0]
    y = torch.empty(n, h, w, cout, dtype=torch.float16, device=x.device)
    ws = _ws(lib().ld_op_conv_skip_ws_bytes(c, sc1, sc2, cout), x.device)
    check(lib().ld_op_conv_skip(_p(x), c, n, h, w, _p(w_packed), _p(bias), _p(s1), sc1, _p(s2), sc2, _p(w_skip), _p(b_skip), _p(rowvec), _p(y), cout,
                                _p(ws), ws.numel(), _stream()), "ld_op_conv_skip")
    return y


def conv2d_gn_partials(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], out_hw: Optional[tuple] = None,
                       residual: Optional[torch.Tensor] = None):
    """3x3 stride-1 NHWC conv that also returns the GroupNorm(32) partial statistics its kernel wrote for the OUTPUT:
    (y, part [n, chunks, 32, 2] fp32 or None when this shape's kernel writes none) — what the executors hand to the next GroupNorm
    (ResBlock1, LD.py:5224-5262; ResnetBlock, LD.py:3560-3576) instead of a statistics pass."""
    import ctypes
    n, h, w, c = x.shape
    hv, wv = (h, w) if out_hw is None else out_hw
    cout = w_packed.shape[0]
    y = torch.empty(n, hv, wv, cout, dtype=torch.float16, device=x.device)
    part = torch.zeros(lib().ld_op_conv_gn_partials_floats(n, hv * wv), dtype=torch.float32, device=x.device)
    chunks = ctypes.c_int(0)
    ws = _ws(192 << 20, x.device)
    check(lib().ld_op_conv_gn_partials(_p(x), c, n, h, w, hv, wv, _p(w_packed), _p(bias), _p(residual), _p(y), cout, _p(part), ctypes.byref(chunks),
                                       _p(ws), ws.numel(), _stream()), "ld_op_conv_gn_partials")
    if chunks.value == 0:
        return y, None
    return y, part[: n * chunks.value * 64].view(n, chunks.value, 32, 2)


def group_norm_silu_conv2d(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, w_packed: torch.Tensor,
                           bias: Optional[torch.Tensor], x2: Optional[torch.Tensor] = None, rowvec: Optional[torch.Tensor] = None,
                           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3(SiLU(GroupNorm32(cat(x, x2)))) + rowvec + residual on NHWC fp16 — ResBlock1.in_layers / out_layers (LD.py:5224-5262)
    as ONE operator: on the halo-tile convolution kernel the normalisation happens inside the conv's A operand."""
    n, h, w, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    cout = w_packed.shape[0]
    y = torch.empty(n, h, w, cout, dtype=torch.float16, device=x.device)
    ws = _ws(lib().ld_op_groupnorm_conv_ws_bytes(c1, c2, n, h, w, cout), x.device)
    check(lib().ld_op_groupnorm_conv(_p(x), c1, _p(x2), c2, n, h, w, _p(gamma), _p(beta), eps, _p(w_packed), _p(bias), _p(rowvec), _p(residual),
                                     _p(y), cout, _p(ws), ws.numel(), _stream()), "ld_op_groupnorm_conv")
    return y


def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, silu: bool = False,
               x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm(32) (+SiLU) over the channel concat of NHWC x and x2, written as one contiguous NHWC tensor."""
    n, c1 = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c1)
    c2 = 0 if x2 is None else x2.shape[-1]
    y = torch.empty(*x.shape[:-1], c1 + c2, dtype=torch.float16, device=x.device)
    ws = _ws(lib().ld_op_groupnorm_ws_bytes(n, hw), x.device)
    check(lib().ld_op_groupnorm(_p(x), c1, _p(x2), c2, n, hw, _p(gamma), _p(beta), eps, int(silu), _p(y), _p(ws), _stream()),
          "ld_op_groupnorm")
    return y


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    c = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().ld_op_layernorm(_p(x), _p(gamma), _p(beta), _p(y), x.numel() // c, c, eps, _stream()), "ld_op_layernorm")
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, causal: bool = False) -> torch.Tensor:
    """optimized_attention(q, k, v, heads) of LD.py:3966-3978: q [b,Lq,heads*d], k/v [b,Lk,heads*d] -> [b,Lq,heads*d].
    (The executor never materialises V^T like this — its projection GEMM writes V^T directly.)"""
    b, lq, c = q.shape
    lk = k.shape[1]
    d = c // heads
    lkp = (lk + 7) // 8 * 8
    vt = torch.zeros(b, c, lkp, dtype=torch.float16, device=q.device)
    vt[:, :, :lk] = v.transpose(1, 2)
    o = torch.empty_like(q)
    check(lib().ld_op_attention(_p(q), c, _p(k.contiguous()), c, _p(vt), lkp, _p(o), c, b, heads, lq, lk, d, 1.0 / math.sqrt(d), int(causal), _stream()),
          "ld_op_attention")
    return o


def attention_qkv(qkv: torch.Tensor, heads: int, causal: bool = False) -> torch.Tensor:
    """Self-attention on a fused projection qkv [b, L, 3*heads*d] = [q | k | v] (how the UNet executor runs attn1 of
    BasicTransformerBlock, LD.py:4117-4162): V is read row-major, no transposed copy."""
    b, l, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    o = torch.empty(b, l, c, dtype=torch.float16, device=qkv.device)
    base = qkv.data_ptr()
    check(lib().ld_op_attention_rowv(base, c3, base + 2 * c, c3, base + 4 * c, c3, _p(o), c, b, heads, l, l, d, 1.0 / math.sqrt(d), int(causal),
                                     _stream()), "ld_op_attention_rowv")
    return o


def attention_rowv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, causal: bool = False) -> torch.Tensor:
    """attention() with V handed over row-major [b, Lk, heads*d]."""
    b, lq, c = q.shape
    lk = k.shape[1]
    d = c // heads
    o = torch.empty_like(q)
    check(lib().ld_op_attention_rowv(_p(q), c, _p(k.contiguous()), c, _p(v.contiguous()), c, _p(o), c, b, heads, lq, lk, d, 1.0 / math.sqrt(d), int(causal),
                                     _stream()), "ld_op_attention_rowv")
    return o


def softmax_rows_(s: torch.Tensor) -> torch.Tensor:
    cols = s.shape[-1]
    check(lib().ld_op_softmax_rows(_p(s), s.numel() // cols, cols, _stream()), "ld_op_softmax_rows")
    return s


def timestep_embed(sigma: torch.Tensor, log_sigmas: torch.Tensor, dim: int):
    n = sigma.numel()
    out = torch.empty(n, dim, dtype=torch.float16, device=sigma.device)
    t = torch.empty(n, dtype=torch.float32, device=sigma.device)
    check(lib().ld_op_timestep_embed(_p(sigma), _p(log_sigmas), log_sigmas.numel(), n, dim, _p(out), _p(t), _stream()),
          "ld_op_timestep_embed")
    return out, t


def cfg_combine(den2: torch.Tensor, cfg: float) -> torch.Tensor:
    """den2 = [uncond ; cond] stacked on dim 0 (fp32) -> uncond + (cond - uncond) * cfg   (cfg_function, LD.py:2594-2606)."""
    half = den2.shape[0] // 2
    out = torch.empty_like(den2[:half])
    check(lib().ld_op_cfg_combine(_p(den2), _p(out), float(cfg), out.numel(), _stream()), "ld_op_cfg_combine")
    return out


def axpby_(x: torch.Tensor, a: float, y: Optional[torch.Tensor] = None, b: float = 0.0, z: Optional[torch.Tensor] = None,
           c: float = 0.0) -> torch.Tensor:
    """x <- a*x + b*y + c*z in place on fp32 latents (the samplers' update arithmetic)."""
    check(lib().ld_op_axpby(_p(x), float(a), _p(y), float(b), _p(z), float(c), x.numel(), _stream()), "ld_op_axpby")
    return x


def bislerp(samples: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """bislerp(samples, width, height) of LD.py:429-518 on the device: [n,c,h,w] -> [n,c,height,width] (fp32 math)."""
    x = samples.float().contiguous()
    n, c, h, w = x.shape
    tmp = torch.empty(n, c, h, width, dtype=torch.float32, device=x.device)
    y = torch.empty(n, c, height, width, dtype=torch.float32, device=x.device)
    check(lib().ld_op_bislerp(_p(x), _p(tmp), _p(y), n, c, h, w, height, width, _stream()), "ld_op_bislerp")
    return y.to(samples.dtype)


# ------------------------------------------------------------------ the `operations=` namespace (secondary seam)
# The reference builds every network from an injectable namespace (`operations=ops`: UNetModel1 LD.py:5338, ResBlock1
# 5207, SpatialTransformer 4179, CrossAttention 4005, FeedForward 3908; BaseModel picks `manual_cast` or
# `disable_weight_init`, LD.py:5809-5816).  The classes below have the torch constructor signatures and state-dict names
# (`weight`, `bias`) of LD.py:2342-2429 and run their forward on the HIP kernels.  Like `disable_weight_init` they do not
# initialise parameters (reset_parameters is a no-op, LD.py:2363); like `manual_cast` they accept any float input dtype
# and return it.  Feature maps keep torch's logical NCHW shape in channels_last memory, which *is* the kernels' NHWC
# layout, so a chain of these modules moves no data between ops.
import torch.nn as nn  # noqa: E402


def _to_f16_cl(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).contiguous(memory_format=torch.channels_last)


class _NoInit:
    def reset_parameters(self):
        return None


class Linear(_NoInit, nn.Module):
    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype), requires_grad=False) if bias else None

    def forward(self, x):
        w = self.weight.to(x.device, torch.float16)
        b = None if self.bias is None else self.bias.to(x.device, torch.float16)
        return linear(x.to(torch.float16).contiguous(), w.contiguous(), b).to(x.dtype)


class Conv2d(_NoInit, nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", device=None, dtype=None):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        st = stride if isinstance(stride, int) else stride[0]
        pd = padding if isinstance(padding, int) else padding[0]
        if k not in (1, 3) or pd != k // 2 or dilation not in (1, (1, 1)) or groups != 1 or padding_mode != "zeros":
            raise NotImplementedError("MI355X Conv2d: kernel 1 or 3 with padding k//2, no dilation / groups (all the SD1.x UNet and VAE use)")
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = in_channels, out_channels, (k, k), (st, st), (pd, pd)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_channels, device=device, dtype=dtype), requires_grad=False) if bias else None
        self._packed = None

    def _weights(self, device):
        key = (self.weight.data_ptr(), self.weight._version, str(device))
        if self._packed is None or self._packed[0] != key:
            wp = repack_conv_weight(self.weight.detach().to(device))
            b = torch.zeros(self.out_channels, dtype=torch.float16, device=device) if self.bias is None else self.bias.detach().to(device, torch.float16)
            self._packed = (key, wp, b.contiguous())
        return self._packed[1], self._packed[2]

    def forward(self, x):
        n, c, h, w = x.shape
        xc = _to_f16_cl(x)
        wp, b = self._weights(x.device)
        y = conv2d(xc.permute(0, 2, 3, 1), wp, b, self.kernel_size[0], self.stride[0])          # a view: channels_last memory is NHWC
        return y.permute(0, 3, 1, 2).to(x.dtype)


class GroupNorm(_NoInit, nn.Module):
    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True, device=None, dtype=None):
        super().__init__()
        if num_groups != 32 or not affine:
            raise NotImplementedError("MI355X GroupNorm: 32 groups, affine (Normalize, LD.py:3931-3939)")
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.empty(num_channels, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(num_channels, device=device, dtype=dtype), requires_grad=False)

    def forward(self, x):
        xc = _to_f16_cl(x) if x.dim() == 4 else x.to(torch.float16).transpose(1, -1).contiguous()
        nhwc = xc.permute(0, 2, 3, 1) if x.dim() == 4 else xc
        y = group_norm(nhwc, self.weight.to(x.device, torch.float16), self.bias.to(x.device, torch.float16), self.eps)
        y = y.permute(0, 3, 1, 2) if x.dim() == 4 else y.transpose(1, -1)
        return y.to(x.dtype)


class LayerNorm(_NoInit, nn.Module):
    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, bias=True, device=None, dtype=None):
        super().__init__()
        c = normalized_shape if isinstance(normalized_shape, int) else normalized_shape[-1]
        if not isinstance(normalized_shape, int) and len(normalized_shape) != 1:
            raise NotImplementedError("MI355X LayerNorm: last-dimension normalisation")
        self.normalized_shape, self.eps = (c,), eps
        self.weight = nn.Parameter(torch.empty(c, device=device, dtype=dtype), requires_grad=False) if elementwise_affine else None
        self.bias = nn.Parameter(torch.empty(c, device=device, dtype=dtype), requires_grad=False) if elementwise_affine and bias else None

    def forward(self, x):
        c = self.normalized_shape[0]
        g = torch.ones(c, dtype=torch.float16, device=x.device) if self.weight is None else self.weight.to(x.device, torch.float16)
        b = torch.zeros(c, dtype=torch.float16, device=x.device) if self.bias is None else self.bias.to(x.device, torch.float16)
        return layer_norm(x.to(torch.float16).contiguous(), g, b, self.eps).to(x.dtype)


def conv_nd(dims, *args, **kwargs):
    """disable_weight_init.conv_nd (LD.py:2407-2412): only dims == 2 exists in the SD1.x networks."""
    if dims == 2:
        return Conv2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def optimized_attention(q, k, v, heads, mask=None):
    """The module-global attention function of LD.py:3966-3988: q [b,Lq,heads*d], k / v [b,Lk,heads*d] -> [b,Lq,heads*d]."""
    if mask is not None:
        raise NotImplementedError("MI355X optimized_attention: the SD1.x UNet passes no mask (LD.py:4028-4036)")
    f16 = lambda t: t.to(torch.float16).contiguous()
    return attention(f16(q), f16(k), f16(v), heads).to(q.dtype)
