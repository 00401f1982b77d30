"""fp32 rounding of Winograd F(6x6,3x3) against F(4x4,3x3) and F(2x2,3x3), emulated in numpy (CPU only, no GPU, no library).

Every transform step is done in float32 in the order the kernels use (FMA chains are emulated as float32 multiply-adds rounded per
operation, which is an upper bound on the fused form's rounding); U = G g G^T is formed in float64 and rounded once, as the packers do.
Reference: direct 3x3 cross-correlation in float64.  Layer shapes are those of the level-3 decoders (models/pwc.lua:76-85: 200 -> 128 ->
128 -> 96 -> 64 -> 32) with Torch's reset() weight scale U(-1/sqrt(9 Ci), 1/sqrt(9 Ci)); activations N(0, s^2) for s = 1e-3 .. 1e3, and a
heavy-tailed variant (5 % of the input channels scaled by 10).

    python tools/wino6_numerics.py            # prints the table kept in profiles/r06_wino6_numerics.txt
"""
import sys
import numpy as np

f32 = np.float32

# interpolation points 0, +-1, +-2, +-1/2, inf (Lavin & Gray; wincnn)
BT6 = np.array([
    [1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0],
    [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0],
    [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
    [0, 1 / 2, 1 / 4, -5 / 2, -5 / 4, 2, 1, 0],
    [0, -1 / 2, 1 / 4, 5 / 2, -5 / 4, -2, 1, 0],
    [0, 2, 4, -5 / 2, -5, 1 / 2, 1, 0],
    [0, -2, 4, 5 / 2, -5, -1 / 2, 1, 0],
    [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]], dtype=np.float64)
G6 = np.array([
    [1, 0, 0],
    [-2 / 9, -2 / 9, -2 / 9],
    [-2 / 9, 2 / 9, -2 / 9],
    [1 / 90, 1 / 45, 2 / 45],
    [1 / 90, -1 / 45, 2 / 45],
    [32 / 45, 16 / 45, 8 / 45],
    [32 / 45, -16 / 45, 8 / 45],
    [0, 0, 1]], dtype=np.float64)
AT6 = np.array([
    [1, 1, 1, 1, 1, 1, 1, 0],
    [0, 1, -1, 2, -2, 1 / 2, -1 / 2, 0],
    [0, 1, 1, 4, 4, 1 / 4, 1 / 4, 0],
    [0, 1, -1, 8, -8, 1 / 8, -1 / 8, 0],
    [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0],
    [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]], dtype=np.float64)

BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)

BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def lin32(M, x, axis):
    """y = M x along `axis` in float32, one rounding per multiply-add, terms in index order (zero coefficients skipped)."""
    x = np.moveaxis(x, axis, 0)
    out = []
    for row in M:
        acc = None
        for c, xi in zip(row, x):
            if c == 0:
                continue
            t = (f32(c) * xi).astype(f32)
            acc = t if acc is None else (acc + t).astype(f32)
        out.append(acc if acc is not None else np.zeros_like(x[0]))
    return np.moveaxis(np.stack(out, 0), 0, axis)


def wino(x, w, BT, G, AT):
    """x: Ci x H x W (float32, H and W multiples of the output tile + 2 halo handled by zero padding), w: Co x Ci x 3 x 3."""
    m = AT.shape[0]
    n = BT.shape[0]
    Ci, H, W = x.shape
    Co = w.shape[0]
    ty, tx = H // m, W // m
    xp = np.zeros((Ci, H + 2, W + 2), f32)
    xp[:, 1:-1, 1:-1] = x
    # tiles: Ci x ty x tx x n x n
    d = np.empty((Ci, ty, tx, n, n), f32)
    for i in range(ty):
        for j in range(tx):
            d[:, i, j] = xp[:, i * m:i * m + n, j * m:j * m + n]
    V = lin32(BT, lin32(BT, d, 3), 4)                               # rows first, then columns (the kernels' order)
    U = np.einsum('ai,ocij,bj->ocab', G, w.astype(np.float64), G).astype(f32)   # double, rounded once
    # M[o, ty, tx, a, b] = sum_c U[o, c, a, b] V[c, ty, tx, a, b]: float32 accumulation in channel order (the MFMA's fmaf chain)
    M = np.zeros((Co, ty, tx, n, n), f32)
    for c in range(Ci):
        M = (M + U[:, c, None, None] * V[None, c]).astype(f32)
    Y = lin32(AT, lin32(AT, M, 4), 3)                               # columns (in registers) first, then rows
    out = np.empty((Co, H, W), f32)
    for i in range(ty):
        for j in range(tx):
            out[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = Y[:, i, j]
    return out


def direct64(x, w):
    Ci, H, W = x.shape
    xp = np.zeros((Ci, H + 2, W + 2), np.float64)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), np.float64)
    for ky in range(3):
        for kx in range(3):
            out += np.einsum('oc,chw->ohw', w[:, :, ky, kx].astype(np.float64), xp[:, ky:ky + H, kx:kx + W])
    return out


def main():
    rng = np.random.default_rng(6)
    H = W = 24                                                       # 24 = lcm-friendly: 12 F(2x2) / 6 F(4x4) / 4 F(6x6) tiles per side
    shapes = [(200, 128), (128, 128), (128, 96), (96, 64), (64, 32), (32, 32)]
    print("error against a float64 direct convolution, normalised by the output's standard deviation: max / mean")
    print("%-10s %-9s | %-21s | %-21s | %-21s | F6/F4 max" % ("layer", "input", "F(2x2)", "F(4x4)", "F(6x6)"))
    for Ci, Co in shapes:
        bound = 1.0 / np.sqrt(9 * Ci)
        w = rng.uniform(-bound, bound, (Co, Ci, 3, 3)).astype(f32)
        for tag in ("s=1e-3", "s=1", "s=1e3", "heavy"):
            x = rng.standard_normal((Ci, H, W)).astype(f32)
            if tag == "heavy":
                hot = rng.random(Ci) < 0.05
                x[hot] *= 10
            elif tag != "s=1":
                x *= f32(float(tag[2:]))
            ref = direct64(x, w)
            sd = ref.std()
            row = []
            for BT, G, AT in ((BT2, G2, AT2), (BT4, G4, AT4), (BT6, G6, AT6)):
                e = np.abs(wino(x, w, BT, G, AT) - ref) / sd
                row.append((e.max(), e.mean()))
            print("%-10s %-9s | %9.2e / %9.2e | %9.2e / %9.2e | %9.2e / %9.2e | %.2f" % (
                "%d->%d" % (Ci, Co), tag, row[0][0], row[0][1], row[1][0], row[1][1], row[2][0], row[2][1], row[2][0] / row[1][0]), flush=True)
    # identity check of the matrices (float64): the three algorithms reproduce the direct form to 1e-12
    x = rng.standard_normal((3, 12, 12))
    w = rng.standard_normal((2, 3, 3, 3))
    ref = direct64(x, w)
    for name, (BT, G, AT) in (("F(6x6)", (BT6, G6, AT6)), ("F(4x4)", (BT4, G4, AT4))):
        m, n = AT.shape
        xp = np.zeros((3, 14, 14))
        xp[:, 1:-1, 1:-1] = x
        out = np.zeros_like(ref)
        for i in range(12 // m):
            for j in range(12 // m):
                d = xp[:, i * m:i * m + n, j * m:j * m + n]
                V = np.einsum('ai,cij,bj->cab', BT, d, BT)
                U = np.einsum('ai,ocij,bj->ocab', G, w, G)
                out[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = np.einsum('ia,oab,jb->oij', AT, np.einsum('ocab,cab->oab', U, V), AT)
        print("%s matrices in float64: max |winograd - direct| = %.1e" % (name, np.abs(out - ref).max()))


if __name__ == "__main__":
    sys.exit(main())
