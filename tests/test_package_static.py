"""CPU: every `alias.name` the package's modules (and bench.py, the tools, the tests' helpers) write against a sibling module of the package
resolves -- `from . import sharded as S ... S.fe_to_bytes_be(...)` on a path that only runs on a GPU box must not be found there first."""
import ast
import glob
import os

import __graft_entry__ as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
zk = G.import_package()
MODULES = {name: getattr(zk, name) for name in ("sumcheck", "gkr", "kzg", "sharded", "mle")}


def aliases_of(tree):
    """{alias: module} for `from . import x as A`, `from zkmle_amd import x as A`, `A = zk.x`, `A = G.import_package().x`"""
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and (node.level == 1 and node.module is None or node.module == "zkmle_amd"):
            for a in node.names:
                if a.name in MODULES:
                    out[a.asname or a.name] = MODULES[a.name]
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and isinstance(node.value, ast.Attribute):
            v = node.value
            if v.attr in MODULES and (isinstance(v.value, ast.Name) and v.value.id == "zk" or isinstance(v.value, ast.Call)):
                out[node.targets[0].id] = MODULES[v.attr]
    return out


def test_sibling_module_attributes_resolve():
    files = glob.glob(os.path.join(ROOT, G.PKG_DIR, "*.py")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    files += glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "_*.py"))
    missing = []
    for f in files:
        tree = ast.parse(open(f).read(), f)
        al = aliases_of(tree)
        seen_alias_modules = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in al:
                mod = al[node.value.id]
                seen_alias_modules.setdefault(node.value.id, set()).add(node.attr)
                if not hasattr(mod, node.attr):
                    missing.append((os.path.relpath(f, ROOT), node.lineno, f"{node.value.id}.{node.attr}", mod.__name__))
    # an alias bound to two different modules in one file (S = zk.sumcheck in one function, S = zk.sharded in another) is judged by the LAST
    # binding above; such files are few and are listed here so that the check stays honest
    assert not missing, missing
