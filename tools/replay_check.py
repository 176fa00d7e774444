import sys, time, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W, ops
from lightdiffusion_amd.unet import synthetic_unet
from lightdiffusion_amd.pipeline import CFGDenoiser
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
unet = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(64, 64))
d = CFGDenoiser(unet, B, 64, 64, 7.0)
d.set_context(torch.randn(1, 77, 768), torch.randn(1, 77, 768))
x = torch.randn(B, 4, 64, 64, device='cuda')
for _ in range(5): d(x, 3.0)
torch.cuda.synchronize()
def t(fn, n=100):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("graph replay only      : %.3f ms" % t(lambda: d._graph.replay()))
print("denoiser call (2 copies + fill + replay): %.3f ms" % t(lambda: d(x, 3.0)))
old = torch.zeros_like(x)
def step():
    den = d(x, 3.0); ops.axpby_(x, 0.99, den, 0.01, old, 0.0); old.copy_(den)
print("full dpmpp-like step   : %.3f ms" % t(step))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); d._graph.replay(); e1.record(); torch.cuda.synchronize()
print("one replay, event-timed: %.3f ms" % e0.elapsed_time(e1))
