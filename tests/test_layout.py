"""Repository layout rules: the product path never imports the oracle (or any CPU fallback), the oracle
says it is test infrastructure, and reference sources are not vendored."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(*dirs):
    for d in dirs:
        for root, _, files in os.walk(os.path.join(REPO, d)):
            for f in files:
                if f.endswith('.py'):
                    yield os.path.join(root, f)


def test_product_never_imports_oracle():
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for path in _py_files('tcvom_amd', 'models'):
        assert not pat.search(open(path).read()), '%s imports the oracle' % path


def test_only_allowed_files_import_oracle():
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    allowed = ('tests' + os.sep, 'bench.py', '__graft_entry__.py', 'oracle' + os.sep)
    for root, dirs, files in os.walk(REPO):
        dirs[:] = [d for d in dirs if d not in ('.git', 'gpurun_out', '__pycache__')]
        for f in files:
            if f.endswith('.py'):
                path = os.path.join(root, f)
                rel = os.path.relpath(path, REPO)
                if pat.search(open(path).read()):
                    assert rel.startswith(allowed), '%s may not import the oracle' % rel


def test_oracle_header_declares_test_infrastructure():
    for path in _py_files('oracle'):
        assert 'TEST INFRASTRUCTURE' in open(path).read(), path


def test_product_has_no_cpu_fallback_switch():
    src = open(os.path.join(REPO, 'tcvom_amd', '_lib.py')).read()
    assert 'no CPU/PyTorch fallback' in src and 'raise ImportError' in src


def test_no_reference_path_at_runtime():
    """Nothing that runs on the GPU box may read /root/reference (only tests/golden/gen_golden.py does)."""
    import ast
    for path in list(_py_files('tcvom_amd', 'models', 'oracle')) + [os.path.join(REPO, 'bench.py'), os.path.join(REPO, '__graft_entry__.py')]:
        tree = ast.parse(open(path).read())
        doc_ids = set()
        for node in ast.walk(tree):                     # docstrings may CITE reference files; code may not use them
            if isinstance(node, (ast.Module, ast.FunctionDef, ast.ClassDef)) and node.body and \
                    isinstance(node.body[0], ast.Expr) and isinstance(node.body[0].value, ast.Constant):
                doc_ids.add(id(node.body[0].value))
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and id(node) not in doc_ids:
                assert '/root/reference' not in node.value, '%s uses the reference tree at run time' % path
