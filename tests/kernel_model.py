"""numpy model of the fast row-scan kernel's indexing (k_interp in
gimp-fix-ca_amd/csrc/fixca_kernels.hip): per-strip 256-column source windows with
edge replication, vertical filter first, horizontal taps gathered by index, fp32
sample arithmetic on fp64 coordinates.  Used on CPU to check that the algorithm
the HIP kernel implements stays within 1 LSB of the oracle and that the planner's
strip width always fits the window."""
import numpy as np


def map_dbl(i, c, size, s, sh):
    d = (i - c) * s + c - sh
    return np.clip(d, 0.0, float(size - 1))


def weights(t, taps):
    t = np.asarray(t, np.float64)
    if taps == 2:
        return np.stack([1 - t, t], -1).astype(np.float32)
    t2 = t * t
    t3 = t2 * t
    return np.stack([(-t3 + 2 * t2 - t) * .5, (3 * t3 - 5 * t2 + 2) * .5,
                     (-3 * t3 + 4 * t2 + t) * .5, (t3 - t2) * .5], -1).astype(np.float32)


def run(img, p, oc, th=32):
    H, W, _ = img.shape
    taps = 4 if p.interpolation == 2 else 2
    back = 1 if taps == 4 else 0
    xc, yc = int(p.lens_x), int(p.lens_y)
    md = max(xc, yc, W - xc, H - yc)
    S = [md / (md + p.red), md / (md + p.blue)]
    XS = [p.x_red, p.x_blue]
    YS = [p.y_red, p.y_blue]
    out = img.copy()
    for c in range(2):
        off = 2 * c
        plane = img[..., off].astype(np.float32)
        for X0 in range(0, W, oc):
            S0 = int(np.floor(map_dbl(min(X0, W - 1), xc, W, S[c], XS[c]))) - back
            cols = np.clip(S0 + np.arange(256), 0, W - 1)
            xs = np.arange(X0, min(X0 + oc, W))
            xd = map_dbl(xs, xc, W, S[c], XS[c])
            x0 = np.floor(xd).astype(int)
            wx = weights(xd - x0, taps)
            t = x0 - back - S0
            assert t.min() >= 0 and t.max() + taps - 1 <= 255, "planner let a window overflow"
            for Y0 in range(0, H, th):
                ys = np.arange(Y0, min(Y0 + th, H))
                yd = map_dbl(ys, yc, H, S[c], YS[c])
                y0 = np.floor(yd).astype(int)
                wy = weights(yd - y0, taps)
                for r, y in enumerate(ys):
                    rows = np.clip(y0[r] - back + np.arange(taps), 0, H - 1)
                    V = np.zeros(256, np.float32)
                    for j in range(taps):
                        V = V + wy[r, j] * plane[rows[j]][cols]
                    acc = np.zeros(len(xs), np.float32)
                    for j in range(taps):
                        acc = acc + wx[:, j] * V[t + j]
                    acc = np.clip(acc, 0, 255) + np.float32(0.5)
                    out[y, xs, off] = acc.astype(np.uint32).astype(np.uint8)
    return out
