"""Dev-container only (skipped wherever /root/reference is absent, i.e. on the GPU box): the committed fixtures under tests/golden/ are what
`python -m oracle.make_golden <every target>` — ONE process, the documented form — produces from the EXECUTED reference modules today, byte
for byte.  This is the pin of the oracle: if the oracle, the generator or a fixture drifts, the md5 comparison says so."""
import hashlib
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGETS = ["sam", "mask_head", "glue", "collate", "preprocess", "dataset", "lisa", "llama_layer", "evaluate"]


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (development container only)")
def test_every_fixture_regenerates_bit_identically(tmp_path):
    work = tmp_path / "repo"
    for d in ("oracle", "medplib_amd", "tests"):
        shutil.copytree(os.path.join(ROOT, d), work / d,
                        ignore=shutil.ignore_patterns("__pycache__", "*.so", "*.o", "obj", ".pytest_cache"))
    before = {f: _md5(work / "tests" / "golden" / f) for f in sorted(os.listdir(work / "tests" / "golden"))}
    r = subprocess.run([sys.executable, "-m", "oracle.make_golden"] + TARGETS, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    after = {f: _md5(work / "tests" / "golden" / f) for f in sorted(os.listdir(work / "tests" / "golden"))}
    assert after == before, {f: (before.get(f), after.get(f)) for f in set(before) | set(after) if before.get(f) != after.get(f)}
    # and the working copy's fixtures are the committed ones
    assert before == {f: _md5(os.path.join(ROOT, "tests", "golden", f)) for f in before}
