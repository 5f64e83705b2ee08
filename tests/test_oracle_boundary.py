"""The oracle is test infrastructure: only tests/, __graft_entry__ (build() compiles the checker, smoke() checks one invocation against
it) and bench.py's cpu_baseline leg may import, call, link or execute anything under oracle/.  A product path that reached the oracle
-- or any CPU fallback -- would void every parity claim, so the rule is a test: every Python file outside tests/ and oracle/ is scanned
for imports of the package, every native source of the product for includes of it."""
import ast
import os
import re

from conftest import ROOT

SKIP_DIRS = {".git", "tests", "oracle", "profiles", "gpurun_out", "__pycache__", "node_modules", "artifacts"}


def _py_files():
    for d, dirs, files in os.walk(ROOT):
        dirs[:] = [x for x in dirs if x not in SKIP_DIRS]
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def _oracle_imports(path):
    """-> [(line, enclosing top-level function or None)] of `import oracle...` / `from oracle... import`"""
    tree = ast.parse(open(path).read(), path)
    out = []

    def visit(node, fn):
        for ch in ast.iter_child_nodes(node):
            f = ch.name if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef)) and fn is None else fn
            if isinstance(ch, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in ch.names):
                out.append((ch.lineno, fn))
            if isinstance(ch, ast.ImportFrom) and (ch.module or "").split(".")[0] == "oracle" and ch.level == 0:
                out.append((ch.lineno, fn))
            visit(ch, f)
    visit(tree, None)
    return out


def test_only_the_checkers_import_the_oracle():
    allowed = {os.path.join(ROOT, "bench.py"): {"cpu_baseline"}, os.path.join(ROOT, "__graft_entry__.py"): {"build", "smoke"}}
    seen = {}
    for p in _py_files():
        imps = _oracle_imports(p)
        if not imps:
            continue
        assert p in allowed, f"{os.path.relpath(p, ROOT)} imports the oracle (lines {[l for l, _ in imps]})"
        for line, fn in imps:
            assert fn in allowed[p], f"{os.path.relpath(p, ROOT)}:{line} imports the oracle outside {sorted(allowed[p])} (in {fn})"
        seen[p] = len(imps)
    assert set(seen) == set(allowed)           # the scan sees the two places that exist: it is not vacuous


def test_no_native_product_source_includes_or_links_the_oracle():
    pat = re.compile(r'#\s*include\s*[<"][^>"]*oracle|-loracle|oracle/_ref|oracle/c/')
    roots = [os.path.join(ROOT, "zk-email-verify_amd"), os.path.join(ROOT, "include"), os.path.join(ROOT, "js")]
    n = 0
    for r in roots:
        for d, dirs, files in os.walk(r):
            dirs[:] = [x for x in dirs if x not in ("node_modules", "__pycache__", "build")]
            for f in files:
                if f.endswith((".h", ".hip", ".c", ".cpp", ".js", ".gyp", "Makefile", ".py")):
                    n += 1
                    for i, line in enumerate(open(os.path.join(d, f), errors="replace"), 1):
                        if line.lstrip().startswith(("//", "*", "/*", "#!")) or (f.endswith(".py") and line.lstrip().startswith("#")):
                            continue
                        assert not pat.search(line), f"{os.path.relpath(os.path.join(d, f), ROOT)}:{i}: {line.strip()[:120]}"
    assert n > 50
