"""Cook-Toom matrices for F(4x4, 3x3) and its fp32 error against F(2x2, 3x3) and the direct form on this model's 3x3 layer shapes (priced, not built).  CPU only."""
import numpy as np
def cook_toom(points, m, r):
    n = m + r - 1
    p = np.array(points, dtype=np.float64)
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for i in range(n - 1):
        Ni = np.prod([p[i] - p[j] for j in range(n - 1) if j != i])
        AT[:, i] = p[i] ** np.arange(m)
        G[i, :] = p[i] ** np.arange(r) / Ni
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    BT = np.zeros((n, n))
    for l in range(n):
        Amat = []; rhs = []
        for k in range(m):
            for j in range(r):
                Amat.append(AT[k, :] * G[:, j]); rhs.append(1.0 if l == k + j else 0.0)
        sol = np.linalg.lstsq(np.array(Amat), np.array(rhs), rcond=None)[0]
        BT[:, l] = sol
    return AT, G, BT
def run(Cin, Cout, Bn, H, variants, seed=0):
    rng = np.random.default_rng(seed)
    r = 3
    x = rng.standard_normal((Bn, H + 2, H + 2, Cin)); x = np.where(x > 0, x, 0.2 * x)
    w = rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)
    ref = np.zeros((Bn, H, H, Cout))
    for a in range(3):
        for b in range(3):
            ref += np.einsum('bhwc,co->bhwo', x[:, a:a+H, b:b+H, :], w[a, b])
    A = np.concatenate([x[:, a:a+H, b:b+H, :].reshape(-1, Cin) for a in range(3) for b in range(3)], 1).astype(np.float32)
    Wm = np.concatenate([w[a, b] for a in range(3) for b in range(3)], 0).astype(np.float32)
    direct = (A @ Wm).reshape(Bn, H, H, Cout)
    out = {'direct': np.abs(direct - ref).max() / np.abs(ref).max()}
    for name, p, mm in variants:
        AT, G, BT = cook_toom(p, mm, r)
        AT32, G32, BT32 = AT.astype(np.float32), G.astype(np.float32), BT.astype(np.float32)
        n = mm + r - 1; nt = H // mm
        d = np.zeros((Bn, nt, nt, n, n, Cin), np.float32); x32 = x.astype(np.float32)
        for ty in range(nt):
            for tx in range(nt):
                d[:, ty, tx] = x32[:, ty*mm:ty*mm+n, tx*mm:tx*mm+n, :]
        V = np.einsum('ia,btsacx->btsicx', BT32, d).astype(np.float32)
        V = np.einsum('jc,btsicx->btsijx', BT32, V).astype(np.float32)
        U = np.einsum('ia,acxo->icxo', G32, w.astype(np.float32)).astype(np.float32)
        U = np.einsum('jc,icxo->ijxo', G32, U).astype(np.float32)
        M = np.zeros((Bn, nt, nt, n, n, Cout), np.float32)
        for i in range(n):
            for j in range(n):
                M[:, :, :, i, j, :] = (V[:, :, :, i, j, :].reshape(-1, Cin) @ U[i, j]).reshape(Bn, nt, nt, Cout)
        Y = np.einsum('ki,btsijo->btskjo', AT32, M).astype(np.float32)
        Y = np.einsum('lj,btskjo->btsklo', AT32, Y).astype(np.float32)
        y = np.zeros((Bn, H, H, Cout), np.float32)
        for ty in range(nt):
            for tx in range(nt):
                y[:, ty*mm:(ty+1)*mm, tx*mm:(tx+1)*mm, :] = Y[:, ty, tx]
        out[name] = np.abs(y - ref).max() / np.abs(ref).max()
    return out
V = [('F2', [0,1,-1], 2), ('F4 {0,1,-1,2,-2}', [0,1,-1,2,-2], 4), ('F4 {0,1,-1,.5,-.5}', [0,1,-1,.5,-.5], 4), ('F4 {0,1,-1,2,-.5}',[0,1,-1,2,-.5],4), ('F4 {0,1,-1,.5,-2}',[0,1,-1,.5,-2],4)]
for Cin, Cout, H in ((256,256,8),(512,512,8),(1152,1024,4),(128,128,16)):
    print(Cin, Cout, H, {k: '%.2e' % v for k, v in run(Cin, Cout, 4, H, V).items()})
np.set_printoptions(linewidth=200, precision=6, suppress=True)
for pts in ([0,1,-1,2,-.5],[0,1,-1,.5,-2],[0,1,-1,2,-2]):
    AT,G,BT = cook_toom(pts,4,3)
    print(pts); print('AT'); print(AT); print('G'); print(G); print('BT'); print(BT)
