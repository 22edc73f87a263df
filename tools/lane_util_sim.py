"""CPU model of the blend kernels' lane utilisation (no GPU): from the ORACLE's state of a view (depth-sorted tile
lists, 2D means, conics, opacities) it replays the per-pixel blend of a sample of tiles with numpy and counts, for the
wave-per-8x8-quadrant mapping the kernels use and for denser mappings (a wave = G lane groups, each a (bw x bh)-pixel
block of the quadrant walking ITS OWN hit list):

  pairs       (block, Gaussian) pairs a wave evaluates (its hit list up to the block's last live pixel)
  members     pairs that contribute to some pixel of the block
  live lanes  (pixel, Gaussian) contributions
  iterations  loop trips of the wave: rounds of 64 list positions, per round the LONGEST hit list among its G groups

usage: python tools/lane_util_sim.py /tmp/state_headline.npz [n_tiles] [seed]
Test infrastructure / design aid: nothing in the product imports it."""
import sys
import numpy as np

LN255 = np.log(255.0)


def rect_min_quadratic(a, b, c, ux0, ux1, uy0, uy1):
    """min over the rectangle [ux0,ux1]x[uy0,uy1] (relative to the centre) of 1/2 (a x^2 + c y^2) + b x y"""
    inside = (ux0 <= 0) & (ux1 >= 0) & (uy0 <= 0) & (uy1 >= 0)

    def q(x, y):
        return 0.5 * (a * x * x + c * y * y) + b * x * y
    rb_c, rb_a = -b / c, -b / a
    m = np.minimum(np.minimum(q(ux0, np.clip(rb_c * ux0, uy0, uy1)), q(ux1, np.clip(rb_c * ux1, uy0, uy1))),
                   np.minimum(q(np.clip(rb_a * uy0, ux0, ux1), uy0), q(np.clip(rb_a * uy1, ux0, ux1), uy1)))
    return np.where(inside, 0.0, m)


def main():
    st = np.load(sys.argv[1])
    n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    W, H = int(st["W"]), int(st["H"])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    m2, co, pl, ranges = st["means2D"].astype(np.float64), st["conic_opacity"].astype(np.float64), st["point_list"], st["ranges"]
    tiles = rng.choice(gx * gy, size=min(n_tiles, gx * gy), replace=False)
    shapes = [(8, 8), (8, 4), (4, 8), (4, 4), (4, 2), (2, 4), (2, 2), (1, 1)]
    acc = {s: dict(pairs_box=0, pairs_ell=0, members=0, iters_box=0, iters_ell=0, ideal_box=0, ideal_ell=0, hyb_box=0, hyb_ell=0, hyb_mem=0) for s in shapes}
    live_total, listed_total, walked_rounds = 0, 0, 0
    for t in tiles:
        tx, ty = t % gx, t // gx
        r0, r1 = ranges[t]
        ids = pl[r0:r1]
        if len(ids) == 0:
            continue
        x, y = m2[ids, 0], m2[ids, 1]
        a, b, c, o = co[ids, 0], co[ids, 1], co[ids, 2], co[ids, 3]
        tau = 1.01 * (LN255 + np.log(np.maximum(o, 1e-30))) + 0.0101
        det = a * c - b * b
        ok = (o * 255.0 > 1.0) & (det > 0)
        hx = np.where(ok, np.sqrt(np.maximum(2 * tau * c / np.maximum(det, 1e-30), 0)), -1.0)
        hy = np.where(ok, np.sqrt(np.maximum(2 * tau * a / np.maximum(det, 1e-30), 0)), -1.0)

        def hits(X0, Y0, w, h, ellipse):
            bx = ok & (x - hx <= X0 + w - 1) & (x + hx >= X0) & (y - hy <= Y0 + h - 1) & (y + hy >= Y0)
            if not ellipse:
                return bx
            qm = rect_min_quadratic(a, b, c, X0 - x, X0 + w - 1 - x, Y0 - y, Y0 + h - 1 - y)
            return bx & (qm <= tau * 1.0001 + 1e-4)
        # this build's tile list: Gaussians whose contribution ellipse reaches the tile (cull_variant 2)
        listed = hits(tx * 16, ty * 16, 16, 16, True)
        ids_l = np.nonzero(listed)[0]
        L = len(ids_l)
        listed_total += L
        if L == 0:
            continue
        x, y, a, b, c, o, hx, hy, tau, ok = (v[ids_l] for v in (x, y, a, b, c, o, hx, hy, tau, ok))
        px = tx * 16 + np.arange(16)[None, :].repeat(16, 0)
        py = ty * 16 + np.arange(16)[:, None].repeat(16, 1)
        inside = (px < W) & (py < H)
        dx = x[:, None, None] - px[None]
        dy = y[:, None, None] - py[None]
        power = -0.5 * (a[:, None, None] * dx * dx + c[:, None, None] * dy * dy) - b[:, None, None] * dx * dy
        alpha = np.minimum(0.99, o[:, None, None] * np.exp(np.minimum(power, 0)))
        valid = (power <= 0) & (alpha >= 1.0 / 255.0) & inside[None]
        cp = np.cumprod(np.where(valid, 1.0 - alpha, 1.0), axis=0)
        stopper = valid & (cp < 1e-4)
        stop = np.where(stopper.any(0), stopper.argmax(0), L)  # position of the entry that ends the pixel (L: never)
        pos = np.arange(L)[:, None, None]
        contrib = valid & (pos < stop[None])
        live_total += int(contrib.sum())
        walk_pix = np.where(inside, np.minimum(stop + 1, L), 0)  # entries a pixel looks at
        for (bw, bh) in shapes:
            A = acc[(bw, bh)]
            for qy in range(2):
                for qx in range(2):
                    QX0, QY0 = tx * 16 + qx * 8, ty * 16 + qy * 8
                    per_group_box, per_group_ell = [], []
                    for by in range(0, 8, bh):
                        for bx_ in range(0, 8, bw):
                            sl = (slice(qy * 8 + by, qy * 8 + by + bh), slice(qx * 8 + bx_, qx * 8 + bx_ + bw))
                            walk = int(walk_pix[sl].max())  # the group stops when its last pixel has stopped
                            if walk == 0:
                                per_group_box.append(np.zeros(0, int)); per_group_ell.append(np.zeros(0, int))
                                continue
                            hb = hits(QX0 + bx_, QY0 + by, bw, bh, False)[:walk]
                            he = hits(QX0 + bx_, QY0 + by, bw, bh, True)[:walk]
                            mem = contrib[(slice(0, walk),) + sl].any(axis=(1, 2))
                            assert not (mem & ~he).any(), "a member failed the (conservative) ellipse test"
                            A["pairs_box"] += int(hb.sum()); A["pairs_ell"] += int(he.sum()); A["members"] += int(mem.sum())
                            per_group_box.append(np.nonzero(hb)[0]); per_group_ell.append(np.nonzero(he)[0])
                    for key, groups in (("box", per_group_box), ("ell", per_group_ell)):
                        n_rounds = (L + 63) // 64
                        cnt = np.zeros((len(groups), n_rounds), int)
                        for g, idx in enumerate(groups):
                            if len(idx):
                                cnt[g] = np.bincount(idx // 64, minlength=n_rounds)
                        A["iters_" + key] += int(cnt.max(0).sum())
                        A["ideal_" + key] += int(cnt.sum())  # / G below
                        # HYBRID: the wave takes its blocks one after the other, G list entries of ONE block per trip
                        # (block pixels x G Gaussians in the 64 lanes): per block and round ceil(hits / G) trips
                        A["hyb_" + key] += int(((cnt + len(groups) - 1) // len(groups)).sum())
            if (bw, bh) == (8, 8):
                walked_rounds += sum((int(walk_pix[qy * 8:qy * 8 + 8, qx * 8:qx * 8 + 8].max()) + 63) // 64 for qy in range(2) for qx in range(2))
    nt = len(tiles)
    print(f"{sys.argv[1]}: {nt} tiles sampled of {gx * gy}; listed entries/tile {listed_total / nt:.1f}; (pixel, Gaussian) contributions/tile {live_total / nt:.0f}; rounds walked/quadrant {walked_rounds / nt / 4:.2f}")
    print("block  G | pairs(box) pairs(ell) members | live/member-pair util | iters(box) iters(ell) ideal(ell) | lane-util of iters(ell) | rel. iters(ell) vs 8x8")
    base = acc[(8, 8)]["iters_ell"]
    print("hybrid (block pixels x G Gaussians per trip, blocks in sequence): trips(box) trips(ell) relative to 8x8 iters(ell)")
    for (bw, bh) in shapes:
        A = acc[(bw, bh)]
        print(f"  {bw}x{bh}: {A['hyb_box'] / nt:9.0f} {A['hyb_ell'] / nt:9.0f}  {A['hyb_box'] / base:.3f} {A['hyb_ell'] / base:.3f}")
    for (bw, bh) in shapes:
        A = acc[(bw, bh)]
        G = 64 // (bw * bh)
        print(f"{bw}x{bh:<2} {G:>3} | {A['pairs_box'] / nt:9.0f} {A['pairs_ell'] / nt:9.0f} {A['members'] / nt:8.0f} | "
              f"{live_total / max(A['members'], 1):6.2f} of {bw * bh:<2} = {live_total / max(A['members'], 1) / (bw * bh):.3f} | "
              f"{A['iters_box'] / nt:9.0f} {A['iters_ell'] / nt:9.0f} {A['ideal_ell'] / G / nt:9.0f} | "
              f"{live_total / max(A['iters_ell'], 1) / 64:.3f} | {A['iters_ell'] / base:.3f}")


if __name__ == "__main__":
    main()
