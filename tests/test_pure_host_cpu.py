"""The pure arithmetic helpers the kernels inline (ihmr_amd/csrc/ihmr_pure.h: point-triangle distance, the +x ray test of a grid column,
the three-instruction division, Rodrigues, the kinematic chain step, the optimizer step) compiled for the HOST by g++ with
-fsanitize=address,undefined and compared bit for bit with the CPU oracle (oracle/sdf_grid.c) and numpy float32 arithmetic, and
against torch where the oracle is torch.  The GPU-less container can therefore unit-test the very functions whose bits the GPU
parity claims rest on; the GPU tests compare the kernels that inline them.  Plus the oracle's own C code under the sanitizers."""
import ctypes
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-fno-fast-math",
       "-march=x86-64-v3"]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("pure")
    exe = str(d / "pure_host_driver")
    subprocess.check_call(["g++"] + SAN + [os.path.join(ROOT, "tests", "pure_host_driver.cpp"), "-o", exe])

    def run(op, records, out_dtype=np.float32):
        records = np.ascontiguousarray(records, np.float32)
        fin, fout = str(d / f"{op}.in"), str(d / f"{op}.out")
        with open(fin, "wb") as fh:
            fh.write(np.int32(records.shape[0]).tobytes())
            fh.write(records.tobytes())
        r = subprocess.run([exe, op, fin, fout], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert r.returncode == 0, r.stderr[-3000:]            # a sanitizer report is a non-zero exit
        return np.fromfile(fout, out_dtype).reshape(records.shape[0], -1)
    return run


def _oracle():
    from oracle import sdf_ref
    return sdf_ref._lib()


def _tris(rng, n, spread=1.0, size=0.3):
    c = rng.uniform(-spread, spread, (n, 1, 3))
    return (c + rng.normal(0, size, (n, 3, 3))).astype(np.float32)


def test_point_triangle_distance_is_the_oracles_bit_for_bit(driver):
    rng = np.random.default_rng(0)
    n = 20000
    tri = _tris(rng, n)
    p = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    # degenerate and special cases: a point ON a corner / edge / face, zero-area and collinear triangles, a duplicated corner
    tri[:50, 1] = tri[:50, 0]
    tri[50:100, 2] = tri[50:100, 0] + 2 * (tri[50:100, 1] - tri[50:100, 0])
    p[100:150] = tri[100:150, 0]
    p[150:200] = 0.5 * (tri[150:200, 0] + tri[150:200, 1])
    p[200:250] = (tri[200:250, 0] + tri[200:250, 1] + tri[200:250, 2]) / 3
    got = driver("ptd", np.concatenate([tri.reshape(n, 9), p], 1))[:, 0]
    L = _oracle()
    ref = np.array([L.ihmr_oracle_point_tri_dist2(tri[i, 0].ctypes.data, tri[i, 1].ctypes.data, tri[i, 2].ctypes.data, p[i].ctypes.data)
                    for i in range(n)], np.float32)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), (int((~same).sum()), got[~same][:4], ref[~same][:4])


def test_column_ray_mask_is_the_oracles_per_voxel_test(driver):
    """sdf_ray_column_hits (one (u, v) test per column + the loop-free t > 0 mask, sdf_ray_hits) against the oracle's per-voxel
    ray_hit_px at all 32 voxel centres of the column -- including near-degenerate triangles (huge 1/det: the mask falls back to the
    per-voxel loop) and triangles that are degenerate in yz (never counted)."""
    rng = np.random.default_rng(1)
    n = 6000
    col = rng.integers(0, 1024, n)
    j, k = col & 31, col >> 5
    py, pz = (2 * j + 1) / 32.0 - 1.0, (2 * k + 1) / 32.0 - 1.0
    tri = np.zeros((n, 3, 3), np.float32)
    tri[:, :, 0] = rng.uniform(-1.2, 1.2, (n, 3))
    tri[:, :, 1] = py[:, None] + rng.normal(0, 0.15, (n, 3))
    tri[:, :, 2] = pz[:, None] + rng.normal(0, 0.15, (n, 3))
    tri[:300, :, 1:] = tri[:300, :1, 1:] + 1e-5 * rng.normal(0, 1, (300, 3, 2))          # slivers in yz: |det| tiny, 1/det huge
    tri[300:400, 2] = tri[300:400, 0]                                                      # degenerate
    tri[400:500, :, 0] = rng.uniform(-1, 1, (100, 1))                                      # planes x = const (crossing exactly between voxels possible)
    out = driver("raycol", np.concatenate([tri.reshape(n, 9), col[:, None].astype(np.float32)], 1), np.uint32)
    L = _oracle()
    px = ((2 * np.arange(32) + 1) / 32.0 - 1.0).astype(np.float32)
    bad = 0
    for i in range(n):
        ref = 0
        p = np.array([0, py[i], pz[i]], np.float32)
        for v in range(32):
            p[0] = px[v]
            ref |= L.ihmr_oracle_ray_hit_px(tri[i, 0].ctypes.data, tri[i, 1].ctypes.data, tri[i, 2].ctypes.data, p.ctypes.data) << v
        bad += int(ref != int(out[i, 2]))
    assert bad == 0, bad
    assert (out[:, 1] == 1).sum() > n // 10 and (out[:, 2] != 0).sum() > n // 20          # (the cases do exercise hits)
    # a restricted `need` mask gives the restriction of the full mask (the static-hand path tests only voxels that are new)
    need = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    rec = np.concatenate([tri.reshape(n, 9), col[:, None].astype(np.float32), need.view(np.float32)[:, None]], 1)
    sub = driver("raycol_need", rec, np.uint32)[:, 0]
    assert np.array_equal(sub, out[:, 2] & need)


def test_three_instruction_division_is_ieee_division(driver):
    rng = np.random.default_rng(2)
    n = 200000
    a = (rng.normal(0, 1, n) * 10.0 ** rng.uniform(-6, 2, n)).astype(np.float32)
    b = (10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    # the plain-division path: divisors outside [1e-6, 1e6] (a degenerate hand keeps the oracle's infinities / NaNs); a non-finite
    # numerator is only meaningful there (the three-instruction form needs finite operands: documented at sdf_div)
    b[:100] = np.float32(1e-7); b[100:220] = np.float32(1e7); b[200:210] = 0.0; a[210:220] = np.inf
    got = driver("div", np.stack([a, b], 1))[:, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = (a / b).astype(np.float32)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), int((~same).sum())


def test_column_range_and_grid_coordinates(driver):
    rng = np.random.default_rng(3)
    n = 5000
    yz = rng.uniform(-1.3, 1.3, (n, 6)).astype(np.float32)
    got = driver("colrange", yz, np.int32)
    centres = (2 * np.arange(32) + 1) / 32.0 - 1.0
    for i in range(0, n, 7):
        y, z = yz[i, :3], yz[i, 3:]
        for lo, hi, (c0, c1) in ((float(y.min()), float(y.max()), got[i, :2]), (float(z.min()), float(z.max()), got[i, 2:])):
            inside = np.nonzero((centres >= lo - 1e-4) & (centres <= hi + 1e-4))[0]
            strict = np.nonzero((centres >= lo) & (centres <= hi))[0]
            have = set(range(c0, c1 + 1))
            assert set(strict) <= have <= set(inside) | {x for x in have if abs(centres[x] - lo) < 2e-4 or abs(centres[x] - hi) < 2e-4}
    x = rng.uniform(-1.5, 1.5, n).astype(np.float32)
    ac = rng.integers(0, 2, n).astype(np.float32)
    out = driver("unnorm", np.stack([x, ac], 1))
    ref = np.where(ac > 0, ((x + np.float32(1)) / np.float32(2)) * np.float32(31), ((x + np.float32(1)) * np.float32(32) - np.float32(1)) / np.float32(2))
    assert np.array_equal(out[:, 0], ref.astype(np.float32))
    ids = np.arange(n) % 32768
    assert np.array_equal(out[:, 1], ((2 * (ids & 31) + 1) / np.float32(32) - np.float32(1)).astype(np.float32))
    assert np.array_equal(out[:, 3], ((2 * (ids >> 10) + 1) / np.float32(32) - np.float32(1)).astype(np.float32))


def test_rodrigues_forward_and_backward_match_the_oracle(driver):
    from oracle.mano_ref import rodrigues_smplx
    rng = np.random.default_rng(4)
    n = 4000
    r = (rng.normal(0, 1, (n, 3)) * rng.uniform(0, 2.5, (n, 1))).astype(np.float32)
    r[:20] = 0.0                                                     # the + 1e-8 branch point
    r[20:40] *= 1e-4
    got = driver("rod_fwd", r).reshape(n, 3, 3)
    rt = torch.tensor(r, dtype=torch.float64, requires_grad=True)
    ref = rodrigues_smplx(rt)
    assert np.abs(got - ref.detach().numpy()).max() < 2e-6
    dR = rng.normal(0, 1, (n, 9)).astype(np.float32)
    (ref * torch.tensor(dR.reshape(n, 3, 3), dtype=torch.float64)).sum().backward()
    gb = driver("rod_bwd", np.concatenate([r, dR], 1))
    big = np.linalg.norm(r, axis=1) > 1e-2                           # (near zero the analytic form divides by the angle: compared where it is conditioned)
    err = np.abs(gb[big] - rt.grad.numpy()[big]).max()
    assert err < 2e-4, err


def test_optimizer_steps_are_torch_optim_in_float32(driver):
    rng = np.random.default_rng(5)
    n = 5000
    x, g = rng.normal(0, 1, n).astype(np.float32), (rng.normal(0, 1, n) * 10.0 ** rng.uniform(-6, 1, n)).astype(np.float32)
    m, v = (0.1 * rng.normal(0, 1, n)).astype(np.float32), (rng.uniform(0, 1, n) ** 4).astype(np.float32)
    t, lr = 7, 1e-2
    step, bc2 = np.float32(lr / (1 - 0.9 ** t)), np.float32(np.sqrt(1 - 0.999 ** t))
    out = driver("adam", np.stack([x, g, m, v, np.full(n, step, np.float32), np.full(n, bc2, np.float32)], 1))
    f = np.float32
    m2 = m + f(0.1) * (g - m)
    v2 = v * f(0.999)
    v2 = v2 + (f(0.001) * g) * g
    x2 = x + (-step) * (m2 / (np.sqrt(v2) / bc2 + f(1e-8)))
    assert np.array_equal(out[:, 1], m2) and np.array_equal(out[:, 2], v2) and np.array_equal(out[:, 0], x2.astype(np.float32))
    p = torch.nn.Parameter(torch.tensor(x))
    opt = torch.optim.Adam([p], lr=lr, betas=(0.9, 0.999))
    p.grad = torch.tensor(g)
    opt.state[p] = dict(step=torch.tensor(float(t - 1)), exp_avg=torch.tensor(m), exp_avg_sq=torch.tensor(v))
    opt.step()
    # (torch's CPU kernels may order the products of addcmul / addcdiv differently: equal to rounding, i.e. relative to the size of the step)
    upd = np.abs(out[:, 0] - x)
    assert np.all(np.abs(out[:, 0] - p.detach().numpy()) <= 2e-4 * upd + 4e-7) and np.abs(out[:, 1] - opt.state[p]["exp_avg"].numpy()).max() < 1e-6
    out = driver("sgd", np.stack([x, g, m, np.full(n, lr, np.float32)], 1))
    mb = m * f(0.9) + g
    assert np.array_equal(out[:, 1], mb) and np.array_equal(out[:, 0], (x + f(-lr) * mb).astype(np.float32))


def test_chain_step_and_loss_helpers(driver):
    rng = np.random.default_rng(6)
    n = 3000
    rec = rng.normal(0, 1, (n, 27)).astype(np.float32)
    out = driver("chain", rec)
    Gp, R, Jj, Jp = rec[:, :12].reshape(n, 3, 4).astype(np.float64), rec[:, 12:21].reshape(n, 3, 3).astype(np.float64), rec[:, 21:24].astype(np.float64), rec[:, 24:].astype(np.float64)
    G = np.concatenate([Gp[:, :, :3] @ R, (Gp[:, :, :3] @ (Jj - Jp)[:, :, None]) + Gp[:, :, 3:]], 2)
    assert np.abs(out[:, :12].reshape(n, 3, 4) - G).max() < 2e-5
    A = G.copy()
    A[:, :, 3] = G[:, :, 3] - (G[:, :, :3] @ Jj[:, :, None])[:, :, 0]
    assert np.abs(out[:, 12:].reshape(n, 3, 4) - A).max() < 5e-5
    ab = rng.normal(0, 1, (n, 6)).astype(np.float32)
    w = rng.choice(np.array([0.0, 1e-8, 0.3, 0.5, 0.51, 1.0], np.float32), n)
    o = driver("misc", np.concatenate([ab, w[:, None]], 1))
    assert np.abs(o[:, :3] - np.cross(ab[:, :3].astype(np.float64), ab[:, 3:].astype(np.float64))).max() < 1e-6
    assert np.array_equal(o[:, 3], np.where(w > 0.5, 0, np.where(w < 1e-7, 21, -1)).astype(np.float32))


def test_oracle_c_code_runs_clean_under_the_sanitizers():
    """`make -C oracle asan`: oracle/sdf_grid.c + its known-answer tests (oracle/sdf_kat.c) as one program under ASan / UBSan."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sdf_kat: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
