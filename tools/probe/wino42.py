"""Cook-Toom matrices for F(m, 2) and the fp32 error of F(4x4, 2x2) against F(2x2, 2x2) and the direct form.  CPU only."""
import numpy as np
np.random.seed(0)
def cook_toom(points, m, r):
    n = m + r - 1
    p = np.array(points, dtype=np.float64)
    assert len(p) == n - 1
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for i in range(n - 1):
        Ni = np.prod([p[i] - p[j] for j in range(n - 1) if j != i])
        AT[:, i] = p[i] ** np.arange(m)
        G[i, :] = p[i] ** np.arange(r) / Ni
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    BT = np.zeros((n, n))
    rows = []; 
    for l in range(n):
        Amat = []; rhs = []
        for k in range(m):
            for j in range(r):
                Amat.append(AT[k, :] * G[:, j]); rhs.append(1.0 if l == k + j else 0.0)
        sol, res, rk, sv = np.linalg.lstsq(np.array(Amat), np.array(rhs), rcond=None)
        BT[:, l] = sol
    return AT, G, BT
def check(AT, G, BT, m, r):
    d = np.random.randn(m + r - 1); g = np.random.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[k + j] * g[j] for j in range(r)) for k in range(m)])
    return np.abs(y - ref).max()
for name, pts, m in (('F(2,2)', [0, 1], 2), ('F(4,2) {0,1,-1,2}', [0, 1, -1, 2], 4), ('F(4,2) {0,1,-1,.5}', [0, 1, -1, 0.5], 4), ('F(4,2) {0,1,-1,-.5}', [0,1,-1,-0.5],4), ('F(3,2) {0,1,-1}', [0,1,-1], 3)):
    AT, G, BT = cook_toom(pts, m, 2)
    print(name, 'identity err', check(AT, G, BT, m, 2))
    print(' AT', np.round(AT, 4).tolist()); print(' G', np.round(G, 4).tolist()); print(' BT', np.round(BT, 4).tolist())

print('---- error simulation ----')
def seq_matmul32(A, B):
    # A [T,K], B [K,N] fp32, sequential accumulation over k (fmaf chain as the MFMA does per k)
    acc = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for k in range(A.shape[1]):
        acc = (acc + A[:, k:k+1] * B[k:k+1, :]).astype(np.float32)   # product rounded then added (not fused): slightly pessimistic
    return acc
def run(Cin, Cout, Bn, H, pts, seed=0):
    rng = np.random.default_rng(seed)
    r = 2
    x = rng.standard_normal((Bn, H + 1, H + 1, Cin))         # one phase sub-input, (H+1)^2 -> H^2 outputs with a 2x2 filter
    x = np.where(x > 0, x, 0.2 * x)
    w = rng.standard_normal((2, 2, Cin, Cout)) / np.sqrt(4 * Cin)
    # f64 reference
    ref = np.zeros((Bn, H, H, Cout))
    for a in range(2):
        for b in range(2):
            ref += np.einsum('bhwc,co->bhwo', x[:, a:a+H, b:b+H, :], w[a, b])
    # direct fp32 sequential over K = 4 Cin
    A = np.concatenate([x[:, a:a+H, b:b+H, :].reshape(-1, Cin) for a in range(2) for b in range(2)], 1).astype(np.float32)
    Wm = np.concatenate([w[a, b] for a in range(2) for b in range(2)], 0).astype(np.float32)
    direct = seq_matmul32(A, Wm).reshape(Bn, H, H, Cout)
    out = {'direct': np.abs(direct - ref).max() / np.abs(ref).max()}
    for name, p, mm in pts:
        AT, G, BT = cook_toom(p, mm, r)
        AT32, G32, BT32 = AT.astype(np.float32), G.astype(np.float32), BT.astype(np.float32)
        n = mm + r - 1
        nt = H // mm
        # tiles
        d = np.zeros((Bn, nt, nt, n, n, Cin), np.float32)
        x32 = x.astype(np.float32)
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
                M[:, :, :, i, j, :] = seq_matmul32(V[:, :, :, i, j, :].reshape(-1, Cin), U[i, j]).reshape(Bn, nt, nt, Cout)
        Y = np.einsum('ki,btsijo->btskjo', AT32, M).astype(np.float32)
        Y = np.einsum('lj,btskjo->btsklo', AT32, Y).astype(np.float32)
        y = np.zeros((Bn, H, H, Cout), np.float32)
        for ty in range(nt):
            for tx in range(nt):
                y[:, ty*mm:(ty+1)*mm, tx*mm:(tx+1)*mm, :] = Y[:, ty, tx]
        out[name] = np.abs(y - ref).max() / np.abs(ref).max()
    return out
pts = [('F(2,2)', [0, 1], 2), ('F(4,2){0,1,-1,2}', [0, 1, -1, 2], 4), ('F(4,2){0,1,-1,.5}', [0, 1, -1, 0.5], 4), ('F(4,2){0,1,-1,-.5}', [0, 1, -1, -0.5], 4)]
for Cin in (256, 512, 1024):
    print(Cin, {k: '%.2e' % v for k, v in run(Cin, 32, 4, 8, pts).items()})
