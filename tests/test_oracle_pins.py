"""The CPU checker is held still: its results on small fixed inputs (tree, hits, a path-traced frame, a
Depth-Box VDB, a marched frame's opacity and sample count) must hash to the committed pins
(tests/golden/oracle_pins.json, written by tests/golden/make_oracle_pins.py). The pins come from this
checker, not from the reference: they pin nothing about parity, they catch an edit of oracle/ that changes
what every GPU test is compared with."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_checker_results_hash_to_the_committed_pins(oracle):
    spec = importlib.util.spec_from_file_location("make_oracle_pins",
                                                  os.path.join(ROOT, "tests", "golden", "make_oracle_pins.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_pins.json")))
    got = mod.compute()
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}
