#!/usr/bin/env python3
"""Workgroup timeline of a bench run from an -DIHMR_TIMELINE build (records written by every workgroup: kind, waves, start, end on the
constant 100 MHz clock, hardware id):   IHMR_HIP_LIBRARY=build/timeline.so IHMR_TIMELINE_OUT=/tmp/tl.npy python3 bench.py ... ;
python3 scripts/timeline_wg.py /tmp/tl.npy
Prints, over the recorded region: mean resident waves per kernel kind and in total (of 256 CUs x 4 SIMDs x 8 wave slots), the share of
the time the chip holds fewer than 25 / 50 / 75 % of the register file (VGPR-weighted residency), and per kernel kind the workgroup
lifetime distribution."""
import sys
import numpy as np

KIND = {1: ("sdf_prep", 64), 2: ("sdf_dist", 128), 3: ("tail<f,f>", 128), 4: ("tail<t,f>", 128), 5: ("tail<t,t>", 128), 6: ("lbs_skin", 128),
        7: ("adam_skel", 64), 8: ("lbs_bwd2", 128), 9: ("lbs_bwd3", 64), 10: ("sample_loss", 128)}
rec = np.load(sys.argv[1])
kind = (rec[:, 0] >> np.uint64(58)).astype(int)
waves = ((rec[:, 0] >> np.uint64(52)) & np.uint64(63)).astype(int)
t0 = (rec[:, 0] & np.uint64((1 << 52) - 1)).astype(np.int64)
t1 = rec[:, 1].astype(np.int64)
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0          # us
T = t1.max()
print(f"{len(rec)} workgroups over {T / 1e3:.2f} ms")
# residency integrals
step = 0.5
nb = int(T / step) + 2
tot_w = np.zeros(nb); tot_v = np.zeros(nb)
print("kind: workgroups, lifetime mean / p50 / p90 / max us, mean resident waves, share of wave-time, share of VGPR-time")
shares = {}
for k in sorted(set(kind)):
    m = kind == k
    name, vg = KIND.get(k, (str(k), 128))
    life = t1[m] - t0[m]
    wt = float((life * waves[m]).sum())
    shares[k] = (wt, wt * vg)
    d = np.zeros(nb)
    np.add.at(d, (t0[m] / step).astype(int), waves[m]); np.add.at(d, (t1[m] / step).astype(int) + 1, -waves[m])
    r = np.cumsum(d)
    tot_w += r; tot_v += r * vg
    print(f"  {name:12s} {m.sum():8d}  {life.mean():6.2f} / {np.percentile(life, 50):6.2f} / {np.percentile(life, 90):6.2f} / {life.max():6.2f}   {wt / T:8.0f}")
sw = sum(v[0] for v in shares.values()); sv = sum(v[1] for v in shares.values())
for k, (a, b) in shares.items():
    print(f"  {KIND.get(k, (str(k),))[0]:12s} wave-time {100 * a / sw:5.1f} %   VGPR-time {100 * b / sv:5.1f} %")
cap_w, cap_v = 256 * 4 * 8, 256 * 4 * 512
print(f"mean resident waves {tot_w.mean():.0f} of {cap_w} slots; mean VGPR residency {100 * tot_v.mean() / cap_v:.1f} % of the register file")
f = tot_v / cap_v
print("register-file residency: " + ", ".join(f"< {int(100 * x)} %: {100 * (f < x).mean():.1f} % of the time" for x in (0.1, 0.25, 0.5, 0.75, 0.9)))

# ---- an excerpt of the residency time series (argv[2] = start ms, default the middle; 2 us per row, 150 rows): VGPR share per kind
if len(sys.argv) > 2 or True:
    start = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else T / 2
    res = 2.0
    rows = 150
    names = [KIND.get(k, (str(k), 128))[0][:9] for k in sorted(set(kind))]
    print("t_us   " + " ".join(f"{n:>9s}" for n in names) + "   total% (of the register file)")
    series = {}
    for k in sorted(set(kind)):
        m = kind == k
        vg = KIND.get(k, (str(k), 128))[1]
        d = np.zeros(rows + 2)
        a = np.clip(((t0[m] - start) / res).astype(int), 0, rows + 1); bb = np.clip(((t1[m] - start) / res).astype(int) + 1, 0, rows + 1)
        live = (t1[m] > start) & (t0[m] < start + rows * res)
        np.add.at(d, a[live], waves[m][live] * vg); np.add.at(d, bb[live], -waves[m][live] * vg)
        series[k] = np.cumsum(d)[:rows] / cap_v * 100
    for r in range(rows):
        vals = [series[k][r] for k in sorted(series)]
        print(f"{r * res:6.0f} " + " ".join(f"{v:9.1f}" for v in vals) + f"   {sum(vals):6.1f}")
