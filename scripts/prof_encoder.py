import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
from ihmr_amd.networks import InterHandEncoder
from ihmr_amd import networks as N
B = 64
enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), torch.zeros(B, 122)).cuda()
img = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
# per-layer timing via monkeypatched conv_igemm
recs = []
orig = N.conv_igemm
def timed(x, pk, Nn, H, W, ldx, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(x, pk, Nn, H, W, ldx, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    Ho, Wo = r[1], r[2]
    fl = 2.0 * Nn * Ho * Wo * pk.cout * pk.kh * pk.kw * pk.cin
    recs.append((dt, fl, f"{pk.kh}x{pk.kw}/{pk.stride} Cin={pk.cin} Cout={pk.cout} HxW={H}x{W}"))
    return r
enc(img); enc(img)
N.conv_igemm = timed
enc(img)
N.conv_igemm = orig
tot = sum(r[0] for r in recs)
agg = {}
for dt, fl, name in recs:
    a = agg.setdefault(name, [0, 0, 0]); a[0] += dt; a[1] += fl; a[2] += 1
for name, (dt, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{dt*1e3:7.3f} ms  x{n}  {fl/dt/1e12:6.1f} TF  {name}")
print("sum conv ms", tot * 1e3)
