"""Repository rules: the product never imports the oracle; HIP sources are gfx950-only."""

import os
import re

from conftest import ROOT


def _py_files(d):
    for dp, _dn, fn in os.walk(d):
        for f in fn:
            if f.endswith(".py"):
                yield os.path.join(dp, f)


def test_product_does_not_import_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for f in _py_files(os.path.join(ROOT, "draco_amd")):
        assert not pat.search(open(f).read()), f"{f} imports the oracle"


def test_no_cuda_dual_path():
    for dp, _dn, fn in os.walk(os.path.join(ROOT, "draco_amd", "csrc")):
        if "build" in dp:
            continue
        for f in fn:
            if f.endswith((".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "__HIP_PLATFORM_AMD__" not in txt and "cuda_runtime" not in txt, f


def test_oracle_headers_say_test_infrastructure():
    for f in _py_files(os.path.join(ROOT, "oracle")):
        head = open(f).read(1500)
        assert "TEST INFRASTRUCTURE ONLY" in head, f
