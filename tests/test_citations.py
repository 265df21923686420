"""Every `path:line` citation of the reference in the boundary headers and documents resolves (three
directory names in include/hns.h did not exist). Runs where the reference checkout is present; a citation is a path
starting with src/, externals/ or Tests/ followed by :line[-line][,line[-line]...]. Bare file names (Kernel.cu:621) are
resolved by a unique basename in the checkout."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FILES = ["include/hns.h", "INTEGRATION.md", "DESIGN.md", "README.md", "oracle/README.md", "oracle/hns_oracle.h", "oracle/hns_oracle.c", "oracle/ref_kernels.cpp",
         "oracle/ref_samplers.cpp"] + sorted(glob.glob(os.path.join(ROOT, "integration", "*.h*"))) + sorted(glob.glob(os.path.join(ROOT, "integration", "*.cpp"))) + \
        sorted(glob.glob(os.path.join(ROOT, "hnanosolver_amd", "host", "*.hpp")))

PATHED = re.compile(r"((?:src|externals|Tests)/[A-Za-z0-9_./+-]+\.(?:cu|cuh|hpp|h|cpp|txt|ds)):(\d+(?:-\d+)?(?:,\s?\d+(?:-\d+)?)*)")
BARE = re.compile(r"(?<![A-Za-z0-9_/.])([A-Z][A-Za-z0-9_]+\.(?:cu|cuh|hpp|h|cpp)):(\d+(?:-\d+)?(?:,\s?\d+(?:-\d+)?)*)")


def _lines(path, cache={}):
    if path not in cache:
        with open(path, errors="replace") as f:
            cache[path] = sum(1 for _ in f)
    return cache[path]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
def test_every_reference_citation_resolves():
    by_name = {}
    for dirpath, _, names in os.walk(REF):
        if "/.git" in dirpath:
            continue
        for n in names:
            by_name.setdefault(n, []).append(os.path.join(dirpath, n))
    bad, n_checked = [], 0
    for rel in FILES:
        path = rel if os.path.isabs(rel) else os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        text = open(path, errors="replace").read()
        found = [(m.group(1), m.group(2), True) for m in PATHED.finditer(text)]
        stripped = PATHED.sub("", text)
        found += [(m.group(1), m.group(2), False) for m in BARE.finditer(stripped)]
        for name, spans, pathed in found:
            if pathed:
                cands = [os.path.join(REF, name)]
                if not os.path.exists(cands[0]):
                    bad.append(f"{os.path.relpath(path, ROOT)}: {name} does not exist in the reference")
                    continue
            else:
                cands = by_name.get(name, [])
                if not cands:
                    continue  # not a reference file (our own sources are cited the same way)
            last = max(int(x) for x in re.findall(r"\d+", spans))
            n_checked += 1
            if not any(_lines(c) >= last for c in cands):
                bad.append(f"{os.path.relpath(path, ROOT)}: {name}:{spans} -- the file has {max(_lines(c) for c in cands)} lines")
    assert n_checked > 100, n_checked
    assert not bad, "\n".join(bad[:40])
