"""VERDICT r5 item 6, step 1 (CPU, numpy): is F(4x4,3x3) accurate enough in fp32 to be worth a kernel?

Error against a float64 direct convolution of: (a) the direct route's arithmetic (an fp32 accumulation chain over K = 9 C
products, two per step like v_mfma_f32_32x32x2_f32), (b) F(2x2,3x3) as csrc/winograd_fused.hip computes it (transforms in fp32,
per-position fp32 chains over C), (c) F(4x4,3x3) with the usual interpolation points (0, +-1, +-2, inf) and (d) with the
better-conditioned set (0, +-1, +-1/2, inf).  Same chain emulation for all of them.  Acceptance (VERDICT): error <= 3 x direct.
Usage: python tools/probes/wino_f4_error.py [C] [Cout] [H] [W]"""
import sys
import numpy as np

f32, f64 = np.float32, np.float64


def chain_matmul(A, B):
    """fp32 [M,K] @ [K,N] accumulated two products per step in a running fp32 sum (MFMA 32x32x2 chain order)"""
    M, K = A.shape
    acc = np.zeros((M, B.shape[1]), f32)
    for k in range(0, K, 2):
        acc = (acc + (A[:, k:k + 1] * B[k:k + 1, :] + A[:, k + 1:k + 2] * B[k + 1:k + 2, :]).astype(f32)).astype(f32)
    return acc


def winograd_matrices(points):
    """Cook-Toom matrices for F(m, 3) from n = m + 2 - 1 finite points + infinity (wincnn construction), float64"""
    import numpy.polynomial.polynomial as P
    n = len(points) + 1          # alpha = m + r - 1
    m = n - 2
    a = np.array(points, f64)
    # AT: m x n, rows i: a_j^i ; last column picks x^(m-1) for the point at infinity
    AT = np.zeros((m, n), f64)
    for i in range(m):
        AT[i, :n - 1] = a ** i
    AT[m - 1, n - 1] = 1.0
    # G: n x 3
    G = np.zeros((n, 3), f64)
    for j in range(n - 1):
        f = np.prod([a[j] - a[k] for k in range(n - 1) if k != j])
        G[j] = np.array([1.0, a[j], a[j] ** 2]) / f
    G[n - 1] = [0, 0, 1.0]
    # BT: n x n, rows are coefficients of the Lagrange-basis numerators  prod_{k != j}(x - a_k)  (and prod_k (x - a_k) for infinity)
    BT = np.zeros((n, n), f64)
    for j in range(n - 1):
        c = np.array([1.0])
        for k in range(n - 1):
            if k != j:
                c = P.polymul(c, np.array([-a[k], 1.0]))
        BT[j, :len(c)] = c
    c = np.array([1.0])
    for k in range(n - 1):
        c = P.polymul(c, np.array([-a[k], 1.0]))
    BT[n - 1, :len(c)] = c
    return AT, G, BT


def check_matrices(AT, G, BT):
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(AT.shape[1]), rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([d[i:i + 3] @ g for i in range(AT.shape[0])])
    assert np.allclose(y, ref, rtol=1e-9, atol=1e-9), (y, ref)


def conv_direct64(x, w):
    C, H, W = x.shape
    O = w.shape[0]
    xp = np.pad(x.astype(f64), ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((O, H, W), f64)
    for r in range(3):
        for s in range(3):
            y += np.einsum("oc,chw->ohw", w[:, :, r, s].astype(f64), xp[:, r:r + H, s:s + W])
    return y


def conv_direct32(x, w):
    C, H, W = x.shape
    O = w.shape[0]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    A = np.stack([xp[:, r:r + H, s:s + W] for r in range(3) for s in range(3)], 0)       # [9, C, H, W]
    A = A.transpose(2, 3, 0, 1).reshape(H * W, 9 * C)                                    # k = (tap, c) like conv_igemm
    Bm = w.transpose(2, 3, 1, 0).reshape(9 * C, O)
    return chain_matmul(A, Bm).reshape(H, W, O).transpose(2, 0, 1)


def conv_winograd32(x, w, AT, G, BT):
    m, n = AT.shape
    C, H, W = x.shape
    O = w.shape[0]
    AT32, G32, BT32 = AT.astype(f32), G.astype(f32), BT.astype(f32)
    U = np.einsum("ij,ocjk,lk->iloc", G32, w, G32).astype(f32)                           # [n, n, O, C] (weights: fp32 like the pack kernels)
    th, tw = -(-H // m), -(-W // m)
    xp = np.zeros((C, th * m + 2, tw * m + 2), f32)
    xp[:, 1:H + 1, 1:W + 1] = x
    tiles = np.stack([xp[:, i * m:i * m + n, j * m:j * m + n] for i in range(th) for j in range(tw)], 0)   # [T, C, n, n]
    # input transform in fp32, one matrix at a time (the kernels add / scale in registers)
    V = np.einsum("ij,tcjk->tcik", BT32, tiles).astype(f32)
    V = np.einsum("tcik,lk->tcil", V, BT32).astype(f32)                                  # [T, C, n, n]
    Mm = np.zeros((n, n, th * tw, O), f32)
    for i in range(n):
        for j in range(n):
            Mm[i, j] = chain_matmul(np.ascontiguousarray(V[:, :, i, j]), np.ascontiguousarray(U[i, j].T))
    Y = np.einsum("ij,jktO->iktO", AT32, Mm).astype(f32)
    Y = np.einsum("iktO,lk->iltO", Y, AT32).astype(f32)                                  # [m, m, T, O]
    y = Y.reshape(m, m, th, tw, O).transpose(4, 2, 0, 3, 1).reshape(O, th * m, tw * m)
    return y[:, :H, :W]


def main():
    C, O, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 256, 16, 32)
    rng = np.random.default_rng(1)
    x = np.maximum(rng.standard_normal((C, H, W)), 0).astype(f32)                        # a ReLU output
    w = (rng.standard_normal((O, C, 3, 3)) * np.sqrt(2.0 / (O * 9))).astype(f32)         # kaiming normal, fan_out (torchvision)
    ref = conv_direct64(x, w)
    scale = np.abs(ref).max()
    out = {}
    out["direct"] = conv_direct32(x, w)
    for name, pts in (("F(2x2)", [0, 1, -1]), ("F(4x4) 0,+-1,+-2", [0, 1, -1, 2, -2]), ("F(4x4) 0,+-1,+-1/2", [0, 1, -1, 0.5, -0.5])):
        AT, G, BT = winograd_matrices(pts)
        check_matrices(AT, G, BT)
        out[name] = conv_winograd32(x, w, AT, G, BT)
    print("C=%d O=%d %dx%d   max|ref|=%.3g" % (C, O, H, W, scale))
    base = None
    for k, v in out.items():
        e = np.abs(v.astype(f64) - ref)
        mx, rms = e.max() / scale, np.sqrt((e ** 2).mean()) / scale
        base = base or (mx, rms)
        print("%-22s max err / max|ref| %.3e (%.2fx direct)   rms %.3e (%.2fx direct)" % (k, mx, mx / base[0], rms, rms / base[1]))


if __name__ == "__main__":
    main()
