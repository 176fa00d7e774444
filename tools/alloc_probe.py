"""Does the speed of the batch-8 forward depend on WHERE the library's allocations land?  (Round 5: separate processes running the same
bench line read 16.3 or 16.6 ms per step, the difference all in conv6_kernel: 6.18 vs 6.40 ms per forward.)  One process, several UNet
instances, a dummy allocation of another size in front of each; prints the addresses (LD_PROFILE_DUMP) and the graph-replayed forward time."""
import os, sys, statistics, random
os.environ['LD_PROFILE_DUMP'] = '1'
import torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet
B = 8
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
keep = []
for it in range(6):
    mb = random.choice([0, 3, 64, 130, 257, 1000])
    if mb:
        keep.append(torch.empty(mb << 20, dtype=torch.uint8, device="cuda"))
    u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(64, 64))
    g = torch.Generator().manual_seed(0)
    u.set_context(torch.randn(2 * B, 77, 768, generator=g))
    x = (torch.randn(B, 4, 64, 64, generator=g) * 3.0).cuda()
    s = torch.full((B,), 3.0, device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        out = u.forward_pair(x, s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            u.forward_pair(x, s, out=out)
        for _ in range(10): gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10): gr.replay()
            e1.record(st); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
    sys.stderr.flush()
    c = u.profile_pair(x, s)
    print(f"instance {it} (dummy {mb} MiB in front): forward {statistics.median(ts):.3f} ms (min {min(ts):.3f});  conv3x3 class {c['conv3x3'][0]:.3f} ms  gemm {c['gemm'][0]:.3f}  attention {c['attention'][0]:.3f}", flush=True)
    del u, gr
    torch.cuda.synchronize()
