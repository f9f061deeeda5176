"""Prototype (numpy, one query at a time) of the Delaunay point location used by k_plane_linear: rows of sorted nodes at
integer heights, a dual-simplex walk on the lifted problem (minimise sum lambda_i |p_i - q|^2 over convex combinations
of sites that reproduce q).  Checked against scipy.interpolate.griddata(method="linear")."""
import sys, os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))


def bary(P, q):
    (x0, y0), (x1, y1), (x2, y2) = P
    d = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
    l1 = ((q[0] - x0) * (y2 - y0) - (x2 - x0) * (q[1] - y0)) / d
    l2 = ((x1 - x0) * (q[1] - y0) - (q[0] - x0) * (y1 - y0)) / d
    return np.array([1 - l1 - l2, l1, l2])


def circum(P):
    (x0, y0), (x1, y1), (x2, y2) = P
    ax, ay, bx, by = x1 - x0, y1 - y0, x2 - x0, y2 - y0
    d = 2 * (ax * by - ay * bx)
    a2, b2 = ax * ax + ay * ay, bx * bx + by * by
    ux, uy = (by * a2 - ay * b2) / d, (ax * b2 - bx * a2) / d
    return x0 + ux, y0 + uy, ux * ux + uy * uy


def locate(rows_x, rows_v, xq, yq, stats=None):
    """rows_x[r]: sorted, strictly increasing valid nodes of row r (height r), rows_v[r] their values.  None = outside."""
    R = len(rows_x)
    # the nearest non-empty rows below / above
    r0 = int(np.floor(yq))
    r0 = min(max(r0, 0), R - 1)
    while r0 >= 0 and len(rows_x[r0]) == 0:
        r0 -= 1
    r1 = int(np.floor(yq)) + 1
    if yq == np.floor(yq) and r0 == yq:  # on a row: any other non-empty row serves as the apex row
        r1 = r0 + 1
    while r1 < R and len(rows_x[r1]) == 0:
        r1 += 1
    if r0 < 0:
        return None
    if r1 >= R:
        if yq != r0:
            return None
        r1 = r0 - 1
        while r1 >= 0 and len(rows_x[r1]) == 0:
            r1 -= 1
        if r1 < 0:
            return None
    A, B = rows_x[r0], rows_x[r1]
    f = (yq - r0) / (r1 - r0)
    # fan start: s(i, j) = (1 - f) A_i + f B_j
    s = lambda i, j: (1 - f) * A[i] + f * B[j]
    if xq < s(0, 0) or xq > s(len(A) - 1, len(B) - 1):
        return None
    pt = lambda rk: (rows_x[rk[0]][rk[1]], float(rk[0]))
    tri = None
    # the start of the kernel (plane.hip): the Delaunay triangulation of the TWO rows alone is known in closed form — the apex
    # of a row's gap is the other row's node nearest to the gap's midpoint — and one of its triangles contains q
    def nearest(X, m):
        k = int(np.searchsorted(X, m, side="left"))
        if k >= len(X):
            return len(X) - 1
        if k > 0 and m - X[k - 1] <= X[k] - m:
            k -= 1
        return k
    iA = int(np.searchsorted(A, xq, side="right") - 1)
    jB = int(np.searchsorted(B, xq, side="right") - 1)
    for k in range(17):
        dk = (k + 1) // 2 if k & 1 else -(k // 2)
        i, j = iA + dk, jB + dk
        if len(A) >= 2 and 0 <= i <= len(A) - 2:
            jj = nearest(B, 0.5 * (A[i] + A[i + 1]))
            if (1 - f) * A[i] + f * B[jj] <= xq <= (1 - f) * A[i + 1] + f * B[jj]:
                cnd = [(r0, i), (r0, i + 1), (r1, jj)]
                if bary([pt(v) for v in cnd], (xq, yq)).min() >= -1e-12:
                    tri = cnd
                    break
        if len(B) >= 2 and 0 <= j <= len(B) - 2:
            ii = nearest(A, 0.5 * (B[j] + B[j + 1]))
            if (1 - f) * A[ii] + f * B[j] <= xq <= (1 - f) * A[ii] + f * B[j + 1]:
                cnd = [(r0, ii), (r1, j), (r1, j + 1)]
                if bary([pt(v) for v in cnd], (xq, yq)).min() >= -1e-12:
                    tri = cnd
                    break
    # else: bracket in both rows
    i = int(np.clip(np.searchsorted(A, xq, side="right") - 1, 0, max(len(A) - 2, 0)))
    j = int(np.clip(np.searchsorted(B, xq, side="right") - 1, 0, max(len(B) - 2, 0)))
    cands = []
    if len(A) >= 2:
        cands += [((r0, i), (r0, i + 1), (r1, j))]
        if len(B) >= 2:
            cands += [((r0, i), (r0, i + 1), (r1, j + 1))]
    if len(B) >= 2:
        cands += [((r0, i), (r1, j), (r1, j + 1))]
        if len(A) >= 2:
            cands += [((r0, i + 1), (r1, j), (r1, j + 1))]
    for cnd in cands:
        if tri is not None:
            break
        lam = bary([pt(v) for v in cnd], (xq, yq))
        if lam.min() >= -1e-12:
            tri = list(cnd)
            break
    if tri is None:  # fan: walk j first, then i
        jj = int(np.searchsorted([s(0, k) for k in range(len(B))], xq, side="right") - 1)
        if jj < len(B) - 1:
            tri = [(r0, 0), (r1, jj), (r1, jj + 1)]
        else:
            ii = int(np.searchsorted([s(k, len(B) - 1) for k in range(len(A))], xq, side="right") - 1)
            ii = min(ii, len(A) - 2)
            tri = [(r0, ii), (r0, ii + 1), (r1, len(B) - 1)]
    npiv = 0
    for it in range(200):
        P = [pt(v) for v in tri]
        cx, cy, R2 = circum(P)
        Rr = np.sqrt(R2)
        best, bestp = None, -1e-10 * max(R2, 1.0)
        for r in range(max(0, int(np.ceil(cy - Rr))), min(R - 1, int(np.floor(cy + Rr))) + 1):
            xs = rows_x[r]
            if len(xs) == 0:
                continue
            w2 = R2 - (r - cy) ** 2
            if w2 <= 0:
                continue
            k = int(np.searchsorted(xs, cx))
            for kk in (k - 1, k):
                if 0 <= kk < len(xs) and (r, kk) not in tri:
                    pw = (xs[kk] - cx) ** 2 - w2
                    if pw < bestp:
                        best, bestp = (r, kk), pw
        if best is None:
            break
        lam = bary(P, (xq, yq))
        mu = bary(P, pt(best))
        ratio = [lam[k] / mu[k] if mu[k] > 1e-14 else np.inf for k in range(3)]
        k = int(np.argmin(ratio))
        tri[k] = best
        npiv += 1
    if stats is not None:
        stats.append(npiv)
    P = [pt(v) for v in tri]
    lam = bary(P, (xq, yq))
    return float(sum(l * rows_v[v[0]][v[1]] for l, v in zip(lam, tri)))


if __name__ == "__main__":
    from scipy.interpolate import griddata

    rng = np.random.default_rng(1)
    worst = 0
    for case in range(40):
        G, nq = (12, int(rng.integers(3, 25))) if case % 2 == 0 else (40, int(rng.integers(3, 12)))
        scale = [0.05, 1.0, 5.0, 30.0][case % 4]
        xs = np.sort(rng.normal(0, scale, (G + 2, nq)) + rng.normal(0, scale, (G + 2, 1)) * 0.3, axis=1)
        if case % 5 == 0:  # precipitation-like: skewed
            xs = np.sort(rng.gamma(0.7, scale * 3, (G + 2, nq)), axis=1)
        vs = rng.normal(0, 1, (G + 2, nq))
        gg = np.repeat(np.arange(G + 2.0)[:, None], nq, 1)
        nqry = 400
        yq = rng.uniform(0.5, G + 0.5, nqry) if case % 3 else rng.integers(1, G + 1, nqry).astype(float)
        xq = rng.uniform(xs.min(), xs.max(), nqry)
        ref = griddata((xs.ravel(), gg.ravel()), vs.ravel(), (xq, yq), method="linear")
        st = []
        rows_x, rows_v = [xs[r] for r in range(G + 2)], [vs[r] for r in range(G + 2)]
        got = np.array([np.nan if (v := locate(rows_x, rows_v, a, b, st)) is None else v for a, b in zip(xq, yq)])
        # ours returns None only outside the strip polygon; scipy NaN outside the hull
        both = ~np.isnan(ref) & ~np.isnan(got)
        err = np.abs(got[both] - ref[both]).max() if both.any() else 0
        worst = max(worst, err)
        print(case, G, nq, scale, "compared", int(both.sum()), "ours-only-nan", int((np.isnan(got) & ~np.isnan(ref)).sum()),
              "ref-only-nan", int((~np.isnan(got) & np.isnan(ref)).sum()), "maxerr %.2e" % err, "pivots mean %.2f max %d" % (np.mean(st), np.max(st)))
    print("worst", worst)
