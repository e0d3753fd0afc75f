"""CPU numerics study (no GPU needed): 1-D Winograd / Toom-Cook F(m, 3) applied to the vocoder's long filters (k = 11: four
3-tap groups accumulated in the Winograd domain) with the library's split-f16 operand scheme (hi + lo f16, three products, fp32
accumulation) -- is the rounding error compatible with the 1e-4 waveform bar?  Input transform in fp32 BEFORE the f16 split,
weights transformed in fp64 at pack time, inverse transform on the fp32 accumulators.  F(3, 3) is the interesting member: its
tile step (3) equals the tap-group width, so one transformed copy of the input serves all groups (F(2,3) / F(4,3) would need
one copy per group phase).  Output: profiles/archive/r02/r02_winograd_numerics.txt; discussion: DESIGN.md section 7."""
import numpy as np
rng = np.random.default_rng(0)
C, Co, L, K = 128, 128, 3072, 11

def split(a):
    hi = a.astype(np.float16); lo = (a - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)

def mm3(Wm, Xm):  # split-f16 3-product contraction, fp32-ish accumulation emulated in float32 matmul
    wh, wl = split(Wm); xh, xl = split(Xm)
    f = lambda a, b: (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float64)
    return f(wh, xh) + f(wh, xl) + f(wl, xh)

x = rng.standard_normal((C, L + K - 1)) * 1.0
x[:, ::97] *= 6  # some outliers
w = rng.standard_normal((Co, C, K)) / np.sqrt(C * K)
# exact
y = np.zeros((Co, L))
for j in range(K):
    y += w[:, :, j] @ x[:, j:j + L]
# direct split-f16
yd = np.zeros((Co, L))
for j in range(K):
    yd += mm3(w[:, :, j] * 64, x[:, j:j + L] * 8) / 512
def rel(a): return np.sqrt(((a - y) ** 2).mean()) / np.sqrt((y ** 2).mean()), np.abs(a - y).max() / np.abs(y).max()
print("direct split-f16      rms rel %.3e  max rel %.3e" % rel(yd))

def wino(points, m):
    """Toom-Cook F(m,3) with the given finite points + infinity."""
    r = 3; n = m + r - 1
    pts = list(points)
    # matrices via Vandermonde construction: Y = A^T [(G g) * (B^T d)]
    import numpy.polynomial.polynomial as P
    # A^T: m x n, rows i: p^i ; last column infinity: only top power
    AT = np.zeros((m, n)); G = np.zeros((n, r)); BT = np.zeros((n, n))
    for k, p in enumerate(pts):
        AT[:, k] = [p ** i for i in range(m)]
        G[k, :] = [p ** i for i in range(r)]
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    # scale G rows by 1/prod(p_k - p_j)
    for k, p in enumerate(pts):
        den = np.prod([p - q for j, q in enumerate(pts) if j != k])
        G[k, :] /= den
    # B^T rows: coefficients of M(x)/(x-p_k) for finite; last row: M(x)
    M = np.array([1.0])
    for q in pts: M = np.convolve(M, np.array([-q, 1.0]))
    for k, p in enumerate(pts):
        poly = np.array([1.0])
        for j, q in enumerate(pts):
            if j != k: poly = np.convolve(poly, np.array([-q, 1.0]))
        BT[k, :len(poly)] = poly
    BT[n - 1, :len(M)] = M
    return AT, G, BT

for name, pts, m in (("F(2,3)", [0, 1, -1], 2), ("F(3,3)", [0, 1, -1, 2], 3), ("F(3,3) pts 0,1,-1,1/2", [0, 1, -1, 0.5], 3),
                     ("F(4,3)", [0, 1, -1, 2, -2], 4), ("F(4,3) pts 0,1,-1,.5,-.5", [0, 1, -1, .5, -.5], 4)):
    AT, G, BT = wino(pts, m)
    n = m + 2
    # sanity on a tiny example
    d = rng.standard_normal(n); g = rng.standard_normal(3)
    ref = np.array([sum(g[j] * d[i + j] for j in range(3)) for i in range(m)])
    assert np.allclose(AT @ ((G @ g) * (BT @ d)), ref, atol=1e-9), name
    Kp = (K + 2) // 3 * 3; ng = Kp // 3
    wp = np.zeros((Co, C, Kp)); wp[:, :, :K] = w
    if m != 3:
        continue_ok = False
    # generic: for group g, input offset 3g; tiles of m outputs at t0 = m*i; need input x[t0+3g : t0+3g+n]
    nt = L // m
    xp = np.zeros((C, L + Kp + n)); xp[:, :x.shape[1]] = x
    yw = np.zeros((Co, nt * m))
    acc = np.zeros((n, Co, nt))
    for gi in range(ng):
        U = np.einsum("kr,ocr->koc", G, wp[:, :, 3 * gi:3 * gi + 3])          # [n][Co][C]
        # input tiles
        idx = (np.arange(nt) * m)[None, :] + 3 * gi + np.arange(n)[:, None]   # [n][nt]
        D = xp[:, idx]                                                         # [C][n][nt]
        V = np.einsum("kn,cnt->kct", BT, D.astype(np.float32).astype(np.float64))  # transform in ~fp32
        V = V.astype(np.float32).astype(np.float64)
        for k in range(n):
            su = max(np.abs(U[k]).max(), 1e-30); sv = 8.0
            acc[k] += mm3(U[k] * (1.0 / su) * 1.0, V[k] * sv) * su / sv
    yw = np.einsum("mk,kot->otm", AT, acc).reshape(Co, nt * m)
    a = yw; b = y[:, :nt * m]
    print("%-28s rms rel %.3e  max rel %.3e   mults/output %.2f vs %d   |BT|max %.1f |G|max %.2f |AT|max %.0f" % (
        name, np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()), np.abs(a - b).max() / np.abs(b).max(),
        ng * n / m, K, np.abs(BT).max(), np.abs(G).max(), np.abs(AT).max()))


# ---- part 2: tile step = group width (F(3,3), F(4,4), F(5,5); F(6,6) end to end: tools/winograd_e2e_numerics.py) for k = 11 and k = 7 --------------------------------------
def split(a):
    hi = a.astype(np.float16); lo = (a - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)
def mm3(Wm, Xm):
    wh, wl = split(Wm); xh, xl = split(Xm)
    f = lambda a, b: (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float64)
    return f(wh, xh) + f(wh, xl) + f(wl, xh)
def toom(points, m, r):
    n = m + r - 1
    pts = list(points); assert len(pts) == n - 1
    AT = np.zeros((m, n)); G = np.zeros((n, r)); BT = np.zeros((n, n))
    for k, p in enumerate(pts):
        AT[:, k] = [p ** i for i in range(m)]
        G[k, :] = [p ** i for i in range(r)]
        G[k, :] /= np.prod([p - q for j, q in enumerate(pts) if j != k])
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    M = np.array([1.0])
    for q in pts: M = np.convolve(M, np.array([-q, 1.0]))
    for k, p in enumerate(pts):
        poly = np.array([1.0])
        for j, q in enumerate(pts):
            if j != k: poly = np.convolve(poly, np.array([-q, 1.0]))
        BT[k, :len(poly)] = poly
    BT[n - 1, :len(M)] = M
    d = rng.standard_normal(n); g = rng.standard_normal(r)
    ref = np.array([sum(g[j] * d[i + j] for j in range(r)) for i in range(m)])
    assert np.allclose(AT @ ((G @ g) * (BT @ d)), ref, atol=1e-8)
    return AT, G, BT
C, Co, L = 128, 128, 3072
for K in (11, 7):
    x = rng.standard_normal((C, L + 64)); x[:, ::97] *= 6
    w = rng.standard_normal((Co, C, K)) / np.sqrt(C * K)
    y = np.zeros((Co, L))
    for j in range(K): y += w[:, :, j] @ x[:, j:j + L]
    for name, pts, m, r in (("F(3,3)", [0, 1, -1, 2], 3, 3), ("F(4,4) 0,+-1,+-2,1/2", [0, 1, -1, 2, -2, 0.5], 4, 4),
                            ("F(4,4) 0,+-1,+-1/2,2", [0, 1, -1, 0.5, -0.5, 2], 4, 4), ("F(2,2)", [0, 1], 2, 2), ("F(5,5)", [0,1,-1,2,-2,.5,-.5,4], 5, 5)):
        AT, G, BT = toom(pts, m, r)
        n = m + r - 1
        Kp = (K + r - 1) // r * r; ng = Kp // r
        wp = np.zeros((Co, C, Kp)); wp[:, :, :K] = w
        nt = L // m
        xp = np.zeros((C, L + Kp + n + 64)); xp[:, :x.shape[1]] = x
        acc = np.zeros((n, Co, nt))
        for gi in range(ng):
            U = np.einsum("kr,ocr->koc", G, wp[:, :, r * gi:r * gi + r])
            idx = (np.arange(nt) * m)[None, :] + r * gi + np.arange(n)[:, None]
            V = np.einsum("kn,cnt->kct", BT, xp[:, idx]).astype(np.float32).astype(np.float64)
            for k in range(n):
                su = max(np.abs(U[k]).max(), 1e-30)
                acc[k] += mm3(U[k] / su, V[k] * 8.0) * su / 8.0
        yw = np.einsum("mk,kot->otm", AT, acc).reshape(Co, nt * m)
        b = y[:, :nt * m]
        print("k=%2d %-24s rms rel %.3e  max rel %.3e   MFMA-mults/output %.2f vs %d (%.2fx)  planes x%.2f" % (
            K, name, np.sqrt(((yw - b) ** 2).mean()) / np.sqrt((b ** 2).mean()), np.abs(yw - b).max() / np.abs(b).max(),
            ng * n / m, K, ng * n / m / K, n / m))
