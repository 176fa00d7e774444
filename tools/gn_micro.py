"""GroupNorm(+SiLU) timing at the UNet's shapes, replayed from a hipGraph; compare with tools/micro/hbm_bw (copy rate)."""
import sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import ops
DEV = "cuda:0"
def case(n, hw, c, reps=20):
    x = torch.randn(n, hw, c, device=DEV, dtype=torch.float16)
    g = torch.ones(c, device=DEV, dtype=torch.float16); b = torch.zeros(c, device=DEV, dtype=torch.float16)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.group_norm(x, g, b, 1e-5, True); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps): ops.group_norm(x, g, b, 1e-5, True)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); gr.replay(); e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = n * hw * c * 2 / 1e6
    print(f"groupnorm n={n} hw={hw} c={c}: {us:7.1f} us  ({mb:.1f} MB tensor; read 2x + write 1x -> {3*mb/us:.2f} TB/s)", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "vae":
    for hw, c in ((262144, 128), (262144, 256), (65536, 256), (65536, 512), (16384, 512), (4096, 512)):
        case(1, hw, c, 10)
    sys.exit(0)
for n in (2, 16):
    for hw, c in ((4096, 320), (4096, 640), (1024, 640), (256, 1280), (64, 1280), (4096, 960)):
        case(n, hw, c)
