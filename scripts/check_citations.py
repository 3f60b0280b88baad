#!/usr/bin/env python3
"""Every `path/file.py:LINE[-LINE][,LINE..]` citation in the header, the kernels, the package, the oracle and the documents must
resolve inside the reference checkout: the file exists and every cited line number lies inside it.  Run where /root/reference is
mounted (the build container); tests/test_abi.py runs it in the CPU suite and skips on machines without the checkout.

usage: python scripts/check_citations.py [--ref /root/reference] [--list]
"""
import argparse
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# a citation = a reference-relative path ending in .py, then one or more line specs; a later bare `:NN-MM` after a comma or
# semicolon continues the same file (the header writes `utils/util_funcs.py:385,420` and `... :400-407`)
PATH = r"(?:[A-Za-z_][\w\-]*/)*[A-Za-z_][\w\-]*\.py"
CITE = re.compile(r"(?<![\w/.])(" + PATH + r"):(\d+(?:\s*-\s*\d+)?(?:\s*,\s*:?\d+(?:\s*-\s*\d+)?)*)")
SOURCES = ["include/*.h", "when-do-gnns-help_amd/csrc/*.hip", "when-do-gnns-help_amd/csrc/*.h", "when-do-gnns-help_amd/*.py",
           "when-do-gnns-help_amd/utils/*.py", "oracle/*.py", "oracle/*.c", "INTEGRATION.md", "DESIGN.md", "bench.py", "__graft_entry__.py"]


def reference_files(ref):
    out = {}
    for dirpath, _dirs, files in os.walk(ref):
        for f in files:
            if f.endswith(".py"):
                full = os.path.join(dirpath, f)
                rel = os.path.relpath(full, ref)
                with open(full, errors="replace") as fh:
                    out[rel] = sum(1 for _ in fh)
    return out


def own_files():
    """paths of THIS repository that look like citations too (tests/test_abi.py:50 ...): not reference citations"""
    own = set()
    for dirpath, _dirs, files in os.walk(ROOT):
        if any(part in dirpath for part in (".git", "gpurun_out", "build", "__pycache__")):
            continue
        for f in files:
            if f.endswith(".py"):
                own.add(os.path.relpath(os.path.join(dirpath, f), ROOT))
    return own


def citations(path):
    text = open(path, errors="replace").read()
    for m in CITE.finditer(text):
        line_no = text.count("\n", 0, m.start()) + 1
        spans = []
        for spec in re.split(r"\s*,\s*:?", m.group(2)):
            nums = [int(x) for x in re.split(r"\s*-\s*", spec.strip()) if x]
            spans.append((nums[0], nums[-1]))
        yield line_no, m.group(1), spans


def check(ref, verbose=False):
    ref_files = reference_files(ref)
    own = own_files()
    own_base = {os.path.basename(p) for p in own}
    bad, n = [], 0
    for pat in SOURCES:
        for path in sorted(glob.glob(os.path.join(ROOT, pat))):
            for line_no, cited, spans in citations(path):
                if cited in ref_files:
                    target = cited
                elif cited in own or (("/" not in cited) and cited in own_base and cited not in {os.path.basename(k) for k in ref_files}):
                    continue  # a path of this repository
                elif "/" not in cited and sum(1 for k in ref_files if os.path.basename(k) == cited) == 1:
                    target = next(k for k in ref_files if os.path.basename(k) == cited)
                elif cited in own_base and cited not in ref_files:
                    continue
                else:
                    bad.append(f"{os.path.relpath(path, ROOT)}:{line_no}: {cited} is not a file of the reference")
                    continue
                n += 1
                for lo, hi in spans:
                    if not (1 <= lo <= hi <= ref_files[target]):
                        bad.append(f"{os.path.relpath(path, ROOT)}:{line_no}: {cited}:{lo}-{hi} lies outside the file ({ref_files[target]} lines)")
                if verbose:
                    print(f"{os.path.relpath(path, ROOT)}:{line_no}: {cited} {spans}")
    return n, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--list", action="store_true")
    a = ap.parse_args()
    if not os.path.isdir(a.ref):
        print(f"{a.ref} is not mounted: nothing to check against")
        return 0
    n, bad = check(a.ref, a.list)
    for b in bad:
        print(b)
    print(f"{n} citations checked, {len(bad)} do not resolve")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
