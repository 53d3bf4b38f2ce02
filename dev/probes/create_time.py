"""Where the creation of the device family (term matrices -> HBM, tile plan) spends its time: cProfile of L.device() at C3"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("WAE_SETUP_DEBUG", "1")
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
L, pb = annulus_family(sys.argv[1] if len(sys.argv) > 1 else "C3", tau=2e-4)
L.device(); L._drop_device()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
L.device()
pr.disable(); print("L.device(): %.3f s" % (time.perf_counter() - t0))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
