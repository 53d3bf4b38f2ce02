"""Local re-discretisation of the tetrahedra around one point: device assembly against numpy formulas, and the
first-order eigenvalue shift both predict (diagnostic for forward_finite_differences_shape_sensitivity)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.sparse as sp
from oracle import fixtures as F
from wae_amd.helmholtz.assemble import assemble_p1

m = np.load(os.path.join(F.GOLDEN_DIR, "rijke_mesh.npz"))
g = np.load(os.path.join(F.GOLDEN_DIR, "rijke_shape.npz"))
t = F.rijke_terms()
w0 = complex(g["omega"][0])
pts, tt, cc = m["points"], m["tetrahedra"].astype(np.int64), m["c_tet"]
n = len(pts)
v = g["v"] / np.sqrt(np.vdot(g["v"], g["v"]))
Lp = 2 * w0 * t["M"] + 1e15 * t["C"]
va = g["v_adj"] / np.conj(np.vdot(g["v_adj"], Lp @ v))


def numpy_local(ph, tets, c):
    X = ph[tets]
    Jm = np.transpose(X[:, :3, :] - X[:, 3:4, :], (0, 2, 1))
    det = np.linalg.det(Jm); Jinv = np.linalg.inv(Jm); adet = np.abs(det)
    ii = np.repeat(tets, 4, axis=1).ravel(); jj = np.tile(tets, (1, 4)).ravel()
    Mloc = (np.ones((4, 4)) + np.eye(4)) / 120.0
    G = np.concatenate([Jinv, -Jinv.sum(axis=1, keepdims=True)], axis=1)
    Mv = (adet[:, None, None] * Mloc).ravel()
    Kv = (-(c ** 2 * adet / 6.0)[:, None, None] * (G @ np.transpose(G, (0, 2, 1)))).ravel()
    return sp.csr_matrix((Mv, (ii, jj)), shape=(n, n)), sp.csr_matrix((Kv, (ii, jj)), shape=(n, n))


p, h = 676, 1e-4
tsel = np.nonzero((tt == p).any(axis=1))[0]
print("adjacent tets", len(tsel), "sign of M diag in fixture", np.sign(t["M"].diagonal()[:3].real), "K diag", np.sign(t["K"].diagonal()[:3].real))
for crd in (1, 2):
    out = {}
    for name, fn in (("numpy", lambda ph: numpy_local(ph, tt[tsel], cc[tsel])),
                     ("device", lambda ph: assemble_p1(ph, tt[tsel].astype(np.int32), cc[tsel]))):
        D = {}
        for sgn in (1, -1):
            ph = pts.copy(); ph[p, crd] += sgn * h
            D[sgn] = fn(ph)
        dM = (D[1][0] - D[-1][0]) / (2 * h); dK = (D[1][1] - D[-1][1]) / (2 * h)
        out[name] = (dM, dK)
        print(crd, name, "sens", -np.vdot(va, (w0 ** 2 * dM + dK) @ v), " parts M:", -np.vdot(va, w0 ** 2 * (dM @ v)), "K:", -np.vdot(va, dK @ v))
    print("   max |dM_dev - dM_np| / max|dM|", abs(out["device"][0] - out["numpy"][0]).max() / abs(out["numpy"][0]).max(),
          " dK:", abs(out["device"][1] - out["numpy"][1]).max() / abs(out["numpy"][1]).max())
