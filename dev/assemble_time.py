#!/usr/bin/env python3
"""Development check: device P1 assembly (wae_p1_assemble) vs the numpy assembly of annulus.build at C2 / C3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz import annulus
from wae_amd.helmholtz.assemble import assemble_p1
for preset in ("C2", "C3"):
    t0 = time.time(); pb = annulus.build(preset); t1 = time.time()
    pts, tets, _ = annulus._mesh(*pb["info"]["grid"])
    ctr = pts[tets].mean(axis=1)
    c_tet = np.where(ctr[:, 2] < annulus.Z_JUMP, annulus.C_COLD, annulus.C_HOT)
    assemble_p1(pts[:100], tets[:1] * 0, None)          # warm-up (library load)
    t2 = time.time(); M, K = assemble_p1(pts, tets, c_tet); t3 = time.time()
    print(preset, "tets", len(tets), "numpy build (M,K,C,Q) %.2f s" % (t1 - t0), " device M,K incl. transfers %.2f s" % (t3 - t2),
          "max rel diff M %.1e K %.1e" % (abs(M - pb["terms"]["M"]).max() / abs(pb["terms"]["M"]).max(), abs(K - pb["terms"]["K"]).max() / abs(pb["terms"]["K"]).max()), flush=True)
