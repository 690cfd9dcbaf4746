"""The work partition of the Stream-K convolution (csrc/encoder.h: conv_streamk_kernel / conv_streamk_fixup_kernel), restated on the host:
worker w owns the K steps [w * T // W, (w + 1) * T // W) of the sequence tile 0 steps 0..nk-1, tile 1 ...; a piece that covers a whole tile is
finished by its worker, any other piece goes to one of the worker's two slots (slot 1: the piece that begins a tile); the fix-up adds the pieces
of a tile in ascending worker (= ascending K) order.  Properties: every step of every tile is owned exactly once; no worker has two pieces in
one slot; the fix-up's walk finds exactly the pieces the workers wrote, in K order; whole tiles are skipped by it."""
import random


def worker_pieces(w, W, tiles, nk):
    total = tiles * nk
    s, s_end = w * total // W, (w + 1) * total // W
    out = []
    while s < s_end:
        tile, kc0 = s // nk, s % nk
        kc1 = min(nk, kc0 + (s_end - s))
        whole = kc0 == 0 and kc1 == nk
        out.append(dict(tile=tile, kc0=kc0, kc1=kc1, whole=whole, slot=None if whole else (1 if kc0 == 0 else 0)))
        s += kc1 - kc0
    return out


def fixup_walk(tile, W, tiles, nk):
    total = tiles * nk
    start = lambda w: w * total // W
    s_lo, s_hi = tile * nk, tile * nk + nk
    w = s_lo * W // total
    while start(w + 1) <= s_lo:
        w += 1
    while start(w) > s_lo:
        w -= 1
    if start(w + 1) >= s_hi:
        return None                      # one worker owns the whole tile
    pieces, ww = [], w
    while ww < W and start(ww) < s_hi:
        pieces.append((ww, 1 if start(ww) <= s_lo else 0))
        ww += 1
    return pieces


def check(W, tiles, nk):
    owned = [[0] * nk for _ in range(tiles)]
    written = {}
    for w in range(W):
        used = set()
        for p in worker_pieces(w, W, tiles, nk):
            for k in range(p["kc0"], p["kc1"]):
                owned[p["tile"]][k] += 1
            if not p["whole"]:
                assert p["slot"] not in used, (W, tiles, nk, w)
                used.add(p["slot"])
                written.setdefault(p["tile"], []).append((w, p["slot"], p["kc0"]))
    assert all(c == 1 for row in owned for c in row), (W, tiles, nk)
    for t in range(tiles):
        walk = fixup_walk(t, W, tiles, nk)
        if walk is None:
            assert t not in written, (W, tiles, nk, t)
        else:
            assert [(w, s) for w, s, _ in written[t]] == walk, (W, tiles, nk, t)
            k0 = [k for _, _, k in written[t]]
            assert k0 == sorted(k0), (W, tiles, nk, t)


def test_resnet50_batch64_layers():
    for tiles, nk in ((196, 144), (392, 72), (100, 288), (196, 64), (100, 128), (392, 64), (400, 64)):   # the layers ihmr_conv_igemm sends to Stream-K
        check(512, tiles, nk)


def test_random_shapes():
    rng = random.Random(7)
    for _ in range(300):
        W = 8 * rng.randint(1, 96)
        tiles, nk = rng.randint(1, 900), rng.randint(1, 300)
        if tiles * nk < W:
            continue
        check(W, tiles, nk)
