"""Repository rules: the oracle is test infrastructure only, and the product has no CPU compute fallback."""
import ast
import os
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _imports(path: Path):
    tree = ast.parse(path.read_text())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom) and node.module:
            yield node.module


def test_product_never_imports_the_oracle():
    for path in (ROOT / "whisperjav_amd").rglob("*.py"):
        for mod in _imports(path):
            assert not mod.split(".")[0] == "oracle", f"{path} imports {mod}"


def test_oracle_importers_are_whitelisted():
    allowed = {"tests", "oracle"}
    for path in ROOT.glob("*.py"):
        uses = [m for m in _imports(path) if m.split(".")[0] == "oracle"]
        if uses:
            assert path.name in ("bench.py", "__graft_entry__.py"), f"{path.name} must not use the oracle"
    # bench.py: only inside the CPU-baseline legs (functions named cpu_baseline*)
    import ast
    tree = ast.parse((ROOT / "bench.py").read_text())
    inside, total = 0, 0
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            names = [node.module or ""] if isinstance(node, ast.ImportFrom) else [a.name for a in node.names]
            total += any(n.split(".")[0] == "oracle" for n in names)
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name.startswith("cpu_baseline")]:
        for node in ast.walk(fn):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                names = [node.module or ""] if isinstance(node, ast.ImportFrom) else [a.name for a in node.names]
                inside += any(n.split(".")[0] == "oracle" for n in names)
    assert total >= 1 and inside == total, f"{total - inside} oracle import(s) of bench.py outside a cpu_baseline* function"


def test_oracle_headers_say_test_infrastructure():
    for path in (ROOT / "oracle").glob("*.py"):
        head = path.read_text()[:1500]
        assert "TEST INFRASTRUCTURE" in head, path


def test_header_cites_reference_call_sites():
    text = (ROOT / "include" / "wjhip.h").read_text()
    for cite in ("faster_whisper_pro_asr.py:819", "whisper_pro_asr.py:433", "silero_v6.py:205-210",
                 "faster_whisper_pro_asr.py:247-253", "device_detector.py"):
        assert cite in text, cite
