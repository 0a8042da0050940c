#!/usr/bin/env python3
"""Takes a case dumped by tools/fuzz_parity.py (LAGO_FUZZ_DUMP) apart: where the scatter-add's error sits, how many terms
pile onto that cell and how large they are (the float32 summation bound eps * sum |terms|)."""
import sys
import numpy as np
z = np.load(sys.argv[1], allow_pickle=True)
name = str(z["name"]); hip, o32, f64 = z["hip"], z["orc32"], z["f64"]
print(name, "shape", hip.shape, "max |f64|", np.abs(f64).max())
eh, eo = np.abs(hip - f64), np.abs(o32 - f64)
ih = np.unravel_index(eh.argmax(), eh.shape)
print("HIP: largest error", eh.max(), "at", ih, " value there", f64[ih], " oracle32 error there", eo[ih], "; oracle32 largest error", eo.max(), "at", np.unravel_index(eo.argmax(), eo.shape))
print("cells with HIP error above 1e-6 of max:", int((eh > 1e-6 * np.abs(f64).max()).sum()), " oracle32:", int((eo > 1e-6 * np.abs(f64).max()).sum()))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import lago_oracle as orc
f8 = lambda a: a.astype(np.float64)
if name.startswith("interp_backward"):
    sabs = orc.interp_backward(np.abs(f8(z["go"])), f8(z["I"]), f8(z["u"]), float(z["dtv"]), True, True)[0]
    cnt = orc.interp_backward(np.ones_like(f8(z["go"])), f8(z["I"]), f8(z["u"]), float(z["dtv"]), True, True)[0]
    print("at HIP's worst cell: sum of |terms| =", sabs[ih], " weight mass (number of samples' worth) =", cnt[ih], " -> eps32 * sum|terms| =", 6e-8 * sabs[ih], "; HIP error / that:", eh[ih] / (6e-8 * sabs[ih]), " oracle32 error / that:", eo[ih] / (6e-8 * sabs[ih]))
elif name.startswith("affine d_I"):
    sabs = orc.affine_interp_backward(np.abs(f8(z["go"])), f8(z["I"]), f8(z["A"]), f8(z["T"]), True, True, True)[0]
    print("at HIP's worst cell: sum of |terms| =", sabs[ih], " -> eps32 * sum|terms| =", 6e-8 * sabs[ih], "; HIP error / that:", eh[ih] / (6e-8 * sabs[ih]), " oracle32:", eo[ih] / (6e-8 * sabs[ih]))
else:
    print("d_A / d_T: a sum over all", z["go"].size, "voxels; sum |go| =", np.abs(z["go"]).sum(), " eps32 * that =", 6e-8 * np.abs(z["go"]).sum())
    print("hip", hip.ravel()[:12]); print("o32", o32.ravel()[:12]); print("f64", f64.ravel()[:12])
