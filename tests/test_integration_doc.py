"""INTEGRATION.md is the reference-side binding a maintainer would paste: its ```python blocks are executed here AS WRITTEN.
CPU: the document's `SpmmJob` mirror against gcc's layout of include/wdg.h (size, every field's offset, every field present);
GPU: the document's own spmm() / spmm_band() on the Cora fixture against the Y the real reference produced."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from _golden import dense_features, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _python_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def _binding_namespace():
    """exec the document's ctypes blocks (those that do not import the package: section 2) in one namespace, from the repo root"""
    blocks = [b for b in _python_blocks() if "wdg_amd" not in b]
    assert len(blocks) >= 2, "INTEGRATION.md lost its binding blocks"
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)  # (the document's library path is relative to the checkout)
    try:
        for b in blocks:
            exec(compile(b, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def _header_struct_fields(name):
    text = open(os.path.join(ROOT, "include", "wdg.h")).read()
    body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        names[0] = names[0].split()[-1]
        fields += [n.strip().lstrip("*") for n in names]
    return fields


def test_module_swap_block_imports():
    """section 1: every name the scripts import exists in the twins (the block is the scripts' import list)"""
    blocks = [b for b in _python_blocks() if "wdg_amd" in b]
    assert blocks
    import wdg_amd  # noqa: F401  (the alias package)
    for b in blocks:
        exec(compile(b, "INTEGRATION.md", "exec"), {})


def test_document_spmm_job_mirror_matches_the_header(tmp_path):
    ns = _binding_namespace()
    mirror = ns["SpmmJob"]
    names = [f for f, _ in mirror._fields_]
    assert names == _header_struct_fields("wdg_spmm_job"), "INTEGRATION.md's SpmmJob and include/wdg.h list different fields"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "wdg.h"', 'int main(void) {',
             'printf("size %zu\\n", sizeof(wdg_spmm_job));']
    lines += [f'printf("{f} %zu\\n", offsetof(wdg_spmm_job, {f}));' for f in names]
    lines += ["return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(got["size"]) == ctypes.sizeof(mirror)
    for f in names:
        assert int(got[f]) == getattr(mirror, f).offset, f
    # and the package's own mirror is the same structure
    import wdg_amd._lib as L
    assert [(f, t) for f, t in L.SpmmJob._fields_] == [(f, t) for f, t in mirror._fields_]


def test_document_band_plan_allocates_what_the_library_writes():
    """the plan kernels write 24 ints of cuts (csrc/spmm_band.hip band_cut_targets): the document must allocate at least that"""
    src = open(os.path.join(ROOT, "when-do-gnns-help_amd", "csrc", "spmm_band.hip")).read()
    written = int(re.search(r"for \(int k = 0; k < (\d+); \+\+k\) cuts\[k\] = 0;", src).group(1))
    block = next(b for b in _python_blocks() if "def band_plan" in b)
    allocated = int(re.search(r"cuts = torch\.empty\((\d+),", block).group(1))
    assert allocated >= written == 24
    assert "n_hub_host" not in open(os.path.join(ROOT, "include", "wdg.h")).read()  # (the plan no longer reads back / syncs)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["rw", "sym"])
def test_document_binding_reproduces_the_reference_on_cora(tag):
    ns = _binding_namespace()
    g0 = load("real_cora")
    n = int(g0["n_nodes"])
    x = torch.from_numpy(dense_features(g0, "featl1_data")).cuda()
    idx = torch.from_numpy(np.stack([g0[f"large_{tag}_row"], g0[f"large_{tag}_col"]]).astype(np.int64))
    adj = torch.sparse_coo_tensor(idx, torch.from_numpy(g0[f"large_{tag}_val"]), (n, n)).cuda()
    gold = g0[f"large_{tag}_y_rows"]
    scale = np.abs(gold).max()
    ys = {}
    for fn in ("spmm", "spmm_band"):
        y = ns[fn](adj, x)
        torch.cuda.synchronize()
        y = y.cpu().numpy()
        # the tolerance is the north star's: 1e-5 on aggregated features
        np.testing.assert_allclose(y[g0["sample_rows"]], gold, rtol=1e-5, atol=1e-6 * scale, err_msg=fn)
        np.testing.assert_allclose(np.linalg.norm(y.astype(np.float64)), g0[f"large_{tag}_y_fro"], rtol=1e-6, err_msg=fn)
        ys[fn] = y
    # both kernel families sum a row's entries in CSR order in fp32 (fused or separate multiply-add): rounding-level agreement
    np.testing.assert_allclose(ys["spmm"], ys["spmm_band"], rtol=2e-6, atol=1e-6 * scale)


def test_document_gnb_job_mirror_matches_the_header(tmp_path):
    ns = _binding_namespace()
    mirror = ns["GnbJob"]
    names = [f for f, _ in mirror._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "wdg.h"', 'int main(void) {', 'printf("size %zu\\n", sizeof(wdg_gnb_job));']
    lines += [f'printf("{f} %zu\\n", offsetof(wdg_gnb_job, {f}));' for f in names]
    lines += ["return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(got["size"]) == ctypes.sizeof(mirror)
    for f in names:
        assert int(got[f]) == getattr(mirror, f).offset, f
    import wdg_amd._lib as L
    assert [(f, t) for f, t in L.GnbJob._fields_] == [(f, t) for f, t in mirror._fields_]


@pytest.mark.gpu
def test_document_gnb_binding_reproduces_scikit_learn_on_texas():
    """the document's gnb_accuracies() on the texas fixture (raw features and a device aggregation of them) against the reference's
    own computation for the branch: GaussianNB().fit(...).predict(...) per epoch and matrix, mean hit rate"""
    from sklearn.naive_bayes import GaussianNB
    ns = _binding_namespace()
    g0 = load("real_texas")
    x = torch.from_numpy(dense_features(g0, "featl1_data")).cuda()
    n = x.shape[0]
    labels = torch.from_numpy(np.asarray(g0["labels"]).reshape(-1).astype(np.int64))
    idx = torch.from_numpy(np.stack([g0["large_rw_row"], g0["large_rw_col"]]).astype(np.int64)) if "large_rw_row" in g0 else None
    if idx is None:
        rng0 = np.random.default_rng(0)
        idx = torch.from_numpy(rng0.integers(0, n, (2, 4 * n)).astype(np.int64))
        adj = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (n, n)).cuda()
    else:
        adj = torch.sparse_coo_tensor(idx, torch.from_numpy(g0["large_rw_val"]), (n, n)).cuda()
    x_agg = ns["spmm"](adj, x)
    rng = np.random.default_rng(4)
    sets = []
    for _ in range(5):
        perm = rng.permutation(n)
        sets.append((torch.from_numpy(np.sort(perm[:110])), torch.from_numpy(np.sort(perm[110:]))))
    c = int(labels.max()) + 1
    got = ns["gnb_accuracies"]([x_agg, x], sets, labels, c).numpy()
    torch.cuda.synchronize()
    lab = labels.numpy()
    for e, (tr, va) in enumerate(sets):
        for k, m in enumerate((x_agg.cpu().numpy(), x.cpu().numpy())):
            want = np.float32(np.mean(GaussianNB().fit(m[tr.numpy()], lab[tr.numpy()]).predict(m[va.numpy()]) == lab[va.numpy()]))
            assert got[e, k] == want, (e, k, got[e, k], want)
