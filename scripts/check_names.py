#!/usr/bin/env python3
"""Every name a function of the package reads from module scope is defined there (or is a builtin): a static stand-in for the
NameError a GPU-only code path would raise on the box.  usage: check_names.py [files ...]  (default: the package)"""
import ast
import builtins
import glob
import os
import symtable
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def module_names(tree, path=None):
    """names bound at module scope (assignments, defs, classes, imports, for / with targets); `from .sibling import *` takes the
    sibling's public names, any other star import is flagged"""
    names, star = set(), False
    for node in tree.body:
        for sub in ast.walk(node) if isinstance(node, (ast.If, ast.Try, ast.For, ast.With)) else [node]:
            if isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                names.add(sub.name)
            elif isinstance(sub, (ast.Import, ast.ImportFrom)):
                for a in sub.names:
                    if a.name == "*":
                        sib = None
                        if isinstance(sub, ast.ImportFrom) and sub.level == 1 and sub.module and path:
                            sib = os.path.join(os.path.dirname(path), sub.module + ".py")
                        if sib and os.path.exists(sib):
                            sib_tree = ast.parse(open(sib).read())
                            got, inner = module_names(sib_tree, sib)
                            exported = [n.value for node in sib_tree.body if isinstance(node, ast.Assign) and
                                        any(isinstance(t, ast.Name) and t.id == "__all__" for t in node.targets)
                                        for n in ast.walk(node.value) if isinstance(n, ast.Constant) and isinstance(n.value, str)]
                            names |= set(exported) if exported else {n for n in got if not n.startswith("_")}
                            star = star or inner
                        else:
                            star = True
                    else:
                        names.add((a.asname or a.name).split(".")[0])
            elif isinstance(sub, (ast.Assign, ast.AugAssign, ast.AnnAssign)):
                targets = sub.targets if isinstance(sub, ast.Assign) else [sub.target]
                for t in targets:
                    for n in ast.walk(t):
                        if isinstance(n, ast.Name):
                            names.add(n.id)
            elif isinstance(sub, (ast.For, ast.With)):
                for n in ast.walk(sub):
                    if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                        names.add(n.id)
    return names, star


def undefined(path):
    src = open(path).read()
    defined, star = module_names(ast.parse(src), path)
    defined |= {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__path__"}
    if star:
        return []
    out = []

    def walk(tab):
        for sym in tab.get_symbols():
            if tab.get_type() != "module" and sym.is_global() and sym.is_referenced():
                if sym.get_name() not in defined and not hasattr(builtins, sym.get_name()):
                    out.append((tab.get_name(), tab.get_lineno(), sym.get_name()))
            elif tab.get_type() == "module" and sym.is_referenced() and not sym.is_assigned() and not sym.is_imported():
                if sym.get_name() not in defined and not hasattr(builtins, sym.get_name()):
                    out.append(("<module>", 0, sym.get_name()))
        for child in tab.get_children():
            walk(child)

    walk(symtable.symtable(src, path, "exec"))
    return out


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "when-do-gnns-help_amd", "**", "*.py"), recursive=True)) + [os.path.join(ROOT, "bench.py")]
    bad = 0
    for f in files:
        for scope, line, name in undefined(f):
            print(f"{os.path.relpath(f, ROOT)}:{line}: {scope} reads undefined global `{name}`")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
