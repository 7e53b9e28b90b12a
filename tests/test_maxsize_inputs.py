"""The inputs behind tests/golden/maxsize.json are rebuilt from the package's generator, not stored: check that the
generator still yields the bytes the oracle digests were taken on (CPU only)."""
import hashlib
import importlib.util
import json
import os

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_generated_inputs_hash_to_the_committed_digests():
    spec = importlib.util.spec_from_file_location("make_maxsize", os.path.join(HERE, "make_maxsize.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(HERE, "maxsize.json")) as f:
        want = json.load(f)
    cases = mod.cases()
    assert sorted(cases) == sorted(want)
    for name, (_, x) in cases.items():
        assert len(x) == want[name]["n"]
        assert hashlib.sha256(x).hexdigest() == want[name]["in_sha256"], name
    assert want["max_input_dict"]["n"] == 1 << 22  # MaxInputSize, lzss/backref.go:10
