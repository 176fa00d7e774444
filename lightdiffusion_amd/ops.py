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


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)


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
    ws = _ws(64 << 20, x.device)
    check(lib().ld_op_conv(_p(x), c1, _p(x2), c2, n, h, w, hv, wv, stride, ksize, _p(w_packed), _p(bias), _p(rowvec), _p(residual),
                           _p(y), cout, _p(ws), ws.numel(), _stream()), "ld_op_conv")
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
