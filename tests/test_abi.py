"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol that
include/waehip.h declares (no compute without a GPU), and the product path fails loudly without a device."""
import os
import re

import numpy as np
import pytest

import wae_amd
from wae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "waehip.h")).read()
    declared = set(re.findall(r"\b(wae_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("wae_family")   # the opaque type
    L = _lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "waehip.h")).read()
    assert "torch" not in hdr.replace("no torch", "") and "at::" not in hdr and "hipStream_t" not in hdr


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import scipy.sparse as sp
    from wae_amd.nlevp import LinearOperatorFamily, Term, pow1
    L = LinearOperatorFamily()
    L.push(Term(sp.identity(4, dtype=complex, format="csr"), (pow1,), (("λ",),), "λ", "I"))
    with pytest.raises(_lib.WaeError):
        L(1.0) @ np.ones(4, dtype=complex)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "wavesandeigenvalues.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f
