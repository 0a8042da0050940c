import os, sys
sys.path.insert(0, os.getcwd())
import torch
import lagomorph_amd as lm
from bench import time_op
ext = lm.lagomorph_ext
lib = ext._lib
met = lm.FluidMetric([0.1, 0.0, 0.01])
for shape, B in (((128,128,128), 32), ((160,160,160), 8), ((160,160,160), 32)):
    m = torch.randn((B, 3) + shape, device="cuda")
    with torch.no_grad():
        ref = met.sharp(m)
        for ipw in (0, 1, 2, 4, 8, 0):
            ext.tune(fluid_xpass_ipw=ipw)
            out = met.sharp(m)
            t, _ = time_op(lambda: met.sharp(m), reps=30, warm=20)
            print(f"{shape} B={B} ipw={ipw}: sharp {t*1e3:7.1f} us  bits {'same' if torch.equal(out, ref) else 'DIFFER'}", flush=True)
    del m, ref, out
    torch.cuda.empty_cache()
