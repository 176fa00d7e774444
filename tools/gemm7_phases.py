"""Where do the intervals of the row-panel kernel (gemm7) go?  A/B build: dbg bit 2048 makes waves 0 and 4 of one workgroup record
s_memtime at phase boundaries of every step; this prints the mean cycles per phase (100 MHz-independent: s_memtime ticks at the shader clock / fixed ref)."""
import os, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
src = open(os.path.join("tools", "gemm5_ab.py")).read().split("cases = [")[0]
exec(src)
cases = [lin(65536, 2560, 320, act=2), lin(65536, 2560, 320), lin(65536, 320, 320, res=True)]
names = ["mfma issue", "-> barrier A", "epi: issue+aux+phase1", "epi: waits", "epi: phase2+stores", "-> barrier B"]
for fn, fl, name in cases:
    for bits in [2048 + int(b) for b in os.environ.get('PH_BITS', '0').split(',')]:
        L.ld_debug_gemm_v5_dbg(bits)
        WS[:4096].zero_()
        fn(); fn(); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); fn(); t1.record(); torch.cuda.synchronize()
        off = 0
        if name.startswith("geglu"):      # ld_op_linear keeps the repacked GEGLU weight + bias at the start of the workspace
            a256 = lambda x: (x + 255) // 256 * 256
            off = a256(2560 * 320 * 2) + a256(2560 * 2)
        ts = WS[off:off + 2 * 32 * 8 * 4].view(torch.int32).cpu().numpy().astype("int64").reshape(2, 32, 8) & 0xffffffff
        print(f"{name}  dbg={bits}  ({t0.elapsed_time(t1) * 1e3:.0f} us per launch incl. launch overhead)")
        nsteps = 32 if "2560" in name else 4
        for g in range(2):
            t = ts[g, :nsteps].copy()
            if name.startswith("geglu"): t[0::2, 4] = t[0::2, 3]      # value steps return after phase (1)
            d = [(t[:, k + 1] - t[:, k]) & 0xffffffff for k in range(6)]
            step = (t[1:, 0] - t[:-1, 0]) & 0xffffffff
            even = [float(x[0::2][1:].mean()) for x in d] if nsteps > 4 else [float(x[0::2].mean()) for x in d]; odd = [float(x[1::2].mean()) for x in d]
            print(f"  group {g}: ticks per step {step.mean():8.0f}   total span {(t[-1, 6] - t[0, 0]) & 0xffffffff}")
            for k in range(6):
                print(f"     {names[k]:24s} even steps {even[k]:8.0f}   odd steps {odd[k]:8.0f}")
    L.ld_debug_gemm_v5_dbg(0)
