"""Experiment: the batch-1 CFG step as TWO concurrent N = 1 forwards (uncond / cond on two streams inside one hipGraph, one UNet replica each)
against the one N = 2 forward.  A batch-1 step is latency-bound (332 dependent launches of ~16 us at 0.1 of the MFMA roof): does concurrency hide it?
Usage: python3 tools/two_stream_b1.py"""
import statistics, sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ctx = torch.randn(2, 77, 768, generator=g)
x = (torch.randn(2, 4, 64, 64, generator=g) * 3.0).cuda()
s = torch.full((2,), 3.0, device="cuda")


def timed(graph, n=40):
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts), min(ts)


# ---- one forward on N = 2
u2 = synthetic_unet(W.sd15_unet_config(), max_batch=2, max_hw=(64, 64))
u2.set_context(ctx)
out2 = torch.empty_like(x)
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    u2.forward(x, s, out=out2)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=st):
        u2.forward(x, s, out=out2)
    print("one N=2 forward:          median %.3f ms  min %.3f ms" % timed(g2), flush=True)

# ---- two concurrent N = 1 forwards, one replica each
ua = synthetic_unet(W.sd15_unet_config(), max_batch=1, max_hw=(64, 64))
ub = synthetic_unet(W.sd15_unet_config(), max_batch=1, max_hw=(64, 64))
ua.set_context(ctx[0:1]); ub.set_context(ctx[1:2])
xa, xb = x[0:1].contiguous(), x[1:2].contiguous()
sa, sb = s[0:1].contiguous(), s[1:2].contiguous()
oa, ob = torch.empty_like(xa), torch.empty_like(xb)
side = torch.cuda.Stream()
with torch.cuda.stream(st):
    ua.forward(xa, sa, out=oa); ub.forward(xb, sb, out=ob)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=st):
        side.wait_stream(st)
        with torch.cuda.stream(side):
            ub.forward(xb, sb, out=ob)
        ua.forward(xa, sa, out=oa)
        st.wait_stream(side)
    print("two concurrent N=1:       median %.3f ms  min %.3f ms" % timed(g1), flush=True)
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gs, stream=st):
        ua.forward(xa, sa, out=oa)
    print("one N=1 forward alone:    median %.3f ms  min %.3f ms" % timed(gs), flush=True)
torch.cuda.synchronize()
err = float((torch.cat([oa, ob]) - out2).abs().max() / out2.abs().max())
print("max relative difference two-stream vs N=2: %.2e" % err)
