#!/usr/bin/env python3
"""Harvests the scene-loader cases of the reference's own XML tests (build container only: reads /root/reference/src/core/tests/test_xml.py).

The reference cannot be imported (no mitsuba / drjit here), but its XML tests only hand STRINGS to `mi.load_string` and state what must
happen: the load succeeds, or it raises with a message matching a pattern.  This script runs the test functions of that file against
recording stand-ins for `mitsuba`, `drjit` and `pytest` -- `mi.load_string(xml, **kwargs)` records its arguments, `pytest.raises(...)` /
`e.match(pattern)` record the expectation -- and writes one record per load:

    {test, line, xml, kwargs, expect: "ok" | "error", pattern: regex or null}

Only VALUES are stored (tests/golden/reference_xml_cases.json): the XML input strings and the expected message patterns -- the inputs and
expected outputs of the reference's tests -- never the text of the test file.  tests/test_reference_xml_cases.py holds the product's
loader and the oracle's independent loader against them.

Usage:  python tests/golden/extract_reference_xml_cases.py [/root/reference]
"""
import inspect
import json
import os
import re
import sys
import types

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "src/core/tests/test_xml.py"

records = []
state = {"test": None, "expect": None}


class _Recorded(Exception):
    pass


def _caller_line():
    for fr in inspect.stack()[2:]:
        if fr.filename == "<reference test_xml.py>":
            return fr.lineno
    return 0


class _Raises:
    """pytest.raises stand-in: the loads made inside are expected to fail; `match=` or a later e.match() gives the message pattern"""

    def __init__(self, exc=Exception, match=None):
        self.pattern = match; self.first = len(records)

    def __enter__(self):
        state["expect"] = self
        return self

    def __exit__(self, et, ev, tb):
        state["expect"] = None
        for r in records[self.first:]:
            r["expect"] = "error"; r["pattern"] = self.pattern
        self.last = len(records)
        return et is None or issubclass(et, _Recorded)

    def match(self, pattern):
        for r in records[self.first:self.last]:
            r["pattern"] = pattern
        return True


def _load_string(xml, **kwargs):
    records.append({"test": state["test"], "line": _caller_line(), "xml": xml, "kwargs": {k: str(v) for k, v in kwargs.items() if k != "parallel"},
                    "expect": "ok", "pattern": None, "call": "load_string"})
    if state["expect"] is not None:
        raise _Recorded()
    return types.SimpleNamespace()


class _Anything:
    """whatever else the tests touch of Mitsuba (logger, file resolver, Scene type): inert"""

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()

    def __eq__(self, other):
        return True


def main():
    text = open(os.path.join(REF, SRC)).read()
    mi = types.ModuleType("mitsuba")
    mi.load_string = _load_string
    for name in ("Thread", "LogLevel", "Scene", "xml", "xml_to_props", "register_bsdf", "BSDF", "load_file", "load_dict"):
        setattr(mi, name, _Anything())
    pt = types.ModuleType("pytest")
    pt.raises = _Raises
    pt.mark = _Anything(); pt.fixture = lambda *a, **k: (lambda f: f); pt.skip = lambda *a, **k: None
    util = types.ModuleType("mitsuba.scalar_rgb.test.util")
    util.fresolver_append_path = lambda f: f
    mods = {"mitsuba": mi, "drjit": _Anything(), "pytest": pt, "mitsuba.scalar_rgb": types.ModuleType("x"), "mitsuba.scalar_rgb.test": types.ModuleType("y"),
            "mitsuba.scalar_rgb.test.util": util}
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        ns = {}
        exec(compile(text, "<reference test_xml.py>", "exec"), ns)
        for name, fn in list(ns.items()):
            if not (name.startswith("test") and callable(fn)):
                continue
            n_args = fn.__code__.co_argcount
            if "tmp_path" in fn.__code__.co_varnames[:n_args] or "xml_to_props" in fn.__code__.co_names or "register_bsdf" in fn.__code__.co_names:
                continue   # mi.xml_to_props / dict_to_xml / Python plugins: facilities outside the scene loader of the hot path
            state["test"] = name
            try:
                fn(*([None] * n_args))
            except Exception as e:   # a test that needs more of Mitsuba than the stand-ins offer: keep what it recorded
                print("  %s stopped: %s" % (name, str(e)[:80]))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    out = {"source": SRC, "what": "inputs and expected outcomes of the reference's XML loader tests (values only)", "records": records}
    path = os.path.join(HERE, "reference_xml_cases.json")
    json.dump(out, open(path, "w"), indent=1)
    n_err = sum(r["expect"] == "error" for r in records)
    print("%d records (%d expect an error, %d with a message pattern) -> %s" % (len(records), n_err, sum(r["pattern"] is not None for r in records), path))


if __name__ == "__main__":
    main()
