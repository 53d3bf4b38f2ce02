"""Prototype of the tile ordering (three nested BFS stages) and window statistics for 256-row tiles."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import breadth_first_order, connected_components
import wae_amd
from wae_amd.helmholtz import annulus

def bfs_levels(G, start):
    order, pred = breadth_first_order(G, start, directed=False, return_predecessors=True)
    lev = np.full(G.shape[0], -1)
    lev[start] = 0
    for v in order[1:]:
        lev[v] = lev[pred[v]] + 1
    return order, lev

def bfs_levels_fast(G, start):
    # level via frontier expansion (vectorised)
    n = G.shape[0]
    lev = np.full(n, -1, dtype=np.int64)
    lev[start] = 0
    frontier = np.array([start])
    l = 0
    indptr, indices = G.indptr, G.indices
    while len(frontier):
        l += 1
        nb = np.unique(np.concatenate([indices[indptr[v]:indptr[v+1]] for v in frontier])) if len(frontier) < 64 else np.unique(indices[np.concatenate([np.arange(indptr[v], indptr[v+1]) for v in frontier])])
        nb = nb[lev[nb] < 0]
        lev[nb] = l
        frontier = nb
    return lev

def levels_sparse(G, start):
    n = G.shape[0]
    lev = np.full(n, -1, dtype=np.int64)
    x = np.zeros(n, dtype=bool); x[start] = True
    lev[start] = 0
    l = 0
    f = x.copy()
    while f.any():
        l += 1
        y = (G @ f.astype(np.int32)) > 0
        y &= lev < 0
        lev[y] = l
        f = y
    return lev

def peripheral(G, comp_nodes_mask=None):
    s = 0
    lev = levels_sparse(G, s)
    for _ in range(2):
        s = int(np.argmax(lev))
        lev = levels_sparse(G, s)
    return s, lev

def order_nodes(G, nodes, depth, thick, out):
    """recursive: nodes = array of global ids forming a subgraph"""
    sub = G[nodes][:, nodes]
    nc, lab = connected_components(sub, directed=False)
    for c in range(nc):
        idx = np.nonzero(lab == c)[0]
        if len(idx) <= 256 or depth == 2:
            if len(idx) == 1:
                out.append(nodes[idx]); continue
            s2 = sub[idx][:, idx]
            st, lev = peripheral(s2)
            o = np.lexsort((idx, lev))          # by level then index
            out.append(nodes[idx[o]])
            continue
        s2 = sub[idx][:, idx]
        st, lev = peripheral(s2)
        nl = lev.max() + 1
        nslab = max(1, int(round(nl / thick)))
        slab = (lev * nslab) // nl
        for sl in range(nslab):
            m = idx[slab == sl]
            if len(m):
                order_nodes(G, nodes[m], depth + 1, thick, out)

def tile_stats(A, perm, TR=256):
    n = A.shape[0]
    Ap = A[perm][:, perm].tocsr()
    W = []
    for t in range(0, n, TR):
        cols = np.unique(Ap.indices[Ap.indptr[t]:Ap.indptr[min(t + TR, n)]])
        W.append(len(cols))
    return np.array(W)

if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "C2"
    thick = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    pb = annulus.build(preset)
    T = pb["terms"]
    A = (abs(T["M"]) + abs(T["K"]) + abs(T["C"]) + abs(T["Q"])).tocsr()
    G = sp.csr_matrix(((A + A.T) != 0).astype(np.int32))
    n = G.shape[0]
    deg = np.diff(G.indptr)
    hub = deg > 8 * np.median(deg)
    print("hubs:", hub.sum())
    Dk = sp.diags((~hub).astype(np.int32))
    G = sp.csr_matrix(Dk @ G @ Dk); G.eliminate_zeros()
    W0 = tile_stats(A, np.arange(n))
    print("lexicographic: window mean %.0f max %d (rows 256)" % (W0.mean(), W0.max()))
    t0 = time.time()
    out = []
    order_nodes(G, np.arange(n), 0, thick, out)
    perm = np.concatenate(out)
    assert len(np.unique(perm)) == n
    print("ordering %.1f s" % (time.time() - t0))
    W = tile_stats(A, perm)
    print("nested BFS thick=%d: window mean %.0f  median %.0f  p90 %.0f max %d" % (thick, W.mean(), np.median(W), np.percentile(W, 90), W.max()))
    np.save("/tmp/perm_%s.npy" % preset, perm)
