"""Repository rules: the oracle is test infrastructure, never product."""
import ast
import os

from conftest import ROOT

PKG = os.path.join(ROOT, "3d_poseestimation_amd")


def _imports(path):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom):
            yield node.module or ""


def test_product_never_imports_oracle_or_reference():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                path = os.path.join(dirpath, f)
                for mod in _imports(path):
                    assert not mod.split(".")[0] == "oracle", f"{path} imports {mod}"
                assert "/root/reference" not in "".join(
                    l for l in open(path) if not l.lstrip().startswith(("#", '"', "/"))
                    and "import" in l), path
    for f in ("bench.py", "__graft_entry__.py"):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            src = open(p).read()
            assert "sys.path.insert(0, \"/root/reference" not in src and "baselineModel" not in src


def test_oracle_headers_say_test_infrastructure():
    for f in os.listdir(os.path.join(ROOT, "oracle")):
        if f.endswith(".py"):
            assert "TEST INFRASTRUCTURE ONLY" in open(os.path.join(ROOT, "oracle", f)).read(), f


def test_no_reference_source_in_repo():
    # fixtures are data: .npz only
    for f in os.listdir(os.path.join(ROOT, "tests", "golden")):
        assert f.endswith(".npz"), f
