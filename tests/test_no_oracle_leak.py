"""The oracle is test infrastructure: the product package must never import, call or read anything under
oracle/, and must not read /root/reference at run time."""
import os
import re

from conftest import ROOT


def test_package_never_touches_oracle_or_reference():
    pkg = os.path.join(ROOT, "omniquant_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "ref_cpu" in src:
                    bad.append(f + ": imports oracle")
                if re.search(r"open\([^)]*/root/reference|sys\.path[^\n]*/root/reference", src):
                    bad.append(f + ": reads /root/reference")
    assert not bad, bad


def test_bench_uses_oracle_only_for_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"oracle", src)]
    assert uses, "bench.py must time the oracle as cpu_baseline"
    fn = src[src.index("def cpu_baseline"):]
    fn = fn[:fn.index("\ndef ", 5)] if "\ndef " in fn[5:] else fn
    outside = src.replace(fn, "")
    assert not re.search(r"^\s*(from|import)\s+oracle\b", outside, flags=re.M)
