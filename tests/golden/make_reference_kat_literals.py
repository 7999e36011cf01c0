#!/usr/bin/env python3
"""Every known-answer LITERAL the reference's own tests assert at the forward boundary, collected without importing or running
anything of the reference: the two test files are parsed with `ast` and the float constants of the known-answer tests
(/root/reference/test/test_esm_sampler.py:269-340, /root/reference/test/test_esm_msa_sampler.py:248-397, 561-565) are listed per
test function.  Output: tests/golden/reference_kat_literals.json (data only: numbers and test names).

tests/test_reference_kats_cpu.py holds tests/golden/reference_kats.json -- the values the checkpoint-gated replays compare against
(tests/test_gpu_esm1.py, tests/test_reference_suite_*.py) -- to this list, so that the day a pretrained checkpoint appears the
replay is known to check every value the reference checks.

  python tests/golden/make_reference_kat_literals.py [/root/reference]   (runs in the build container only: the reference never travels)
"""
import ast
import json
import os
import sys

FILES = (("test/test_esm_sampler.py", 260, 345), ("test/test_esm_msa_sampler.py", 240, 570))


def _is_kat(v):
    """A known-answer float: a log-likelihood with (nearly) full float32 print precision, not a tolerance or a temperature."""
    if not isinstance(v, float):
        return False
    digits = len(repr(abs(v)).replace(".", "").lstrip("0"))
    return 0.01 < abs(v) < 20.0 and digits >= 8


def collect(ref_root):
    out = {}
    for rel, lo, hi in FILES:
        tree = ast.parse(open(os.path.join(ref_root, rel)).read())
        per = {}
        for node in tree.body:
            if not isinstance(node, ast.FunctionDef) or node.end_lineno < lo or node.lineno > hi:
                continue
            vals = set()
            # the function body and its decorators (pytest.mark.parametrize tables hold most of the values)
            for sub in ast.walk(node):
                if isinstance(sub, ast.Constant) and _is_kat(sub.value):
                    # a unary minus is a separate node: look the sign up in the source segment
                    vals.add(sub.value)
            for sub in ast.walk(node):
                if isinstance(sub, ast.UnaryOp) and isinstance(sub.op, ast.USub) and isinstance(sub.operand, ast.Constant) \
                        and _is_kat(sub.operand.value):
                    vals.discard(sub.operand.value)
                    vals.add(-sub.operand.value)
            if vals:
                per[node.name] = sorted(vals)
        out[rel] = per
    return out


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    doc = {"_source": "float literals of the reference's known-answer tests, parsed with ast (nothing imported or executed): "
                      + ", ".join("%s:%d-%d" % f for f in FILES), "files": collect(ref_root)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kat_literals.json")
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    n = sum(len(v) for f in doc["files"].values() for v in f.values())
    print("wrote %s: %d literals in %d test functions" % (path, n, sum(len(f) for f in doc["files"].values())))


if __name__ == "__main__":
    main()
