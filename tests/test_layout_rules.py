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
    text = (ROOT / "bench.py").read_text()
    assert text.count("from oracle import") == 1 and "def cpu_baseline" in text   # only inside the CPU-baseline leg


def test_oracle_headers_say_test_infrastructure():
    for path in (ROOT / "oracle").glob("*.py"):
        head = path.read_text()[:1500]
        assert "TEST INFRASTRUCTURE" in head, path


def test_header_cites_reference_call_sites():
    text = (ROOT / "include" / "wjhip.h").read_text()
    for cite in ("faster_whisper_pro_asr.py:819", "whisper_pro_asr.py:433", "silero_v6.py:205-210",
                 "faster_whisper_pro_asr.py:247-253", "device_detector.py"):
        assert cite in text, cite
