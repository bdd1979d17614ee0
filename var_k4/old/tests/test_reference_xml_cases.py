"""The scene loaders against the reference's OWN XML tests (src/core/tests/test_xml.py; inputs and expected outcomes harvested by
tests/golden/extract_reference_xml_cases.py into tests/golden/reference_xml_cases.json): every record is an XML string the reference's test
hands to mi.load_string together with what must happen -- the load succeeds, or it fails with a message matching the test's pattern.  Held
against the product's loader (scene_loader.cpp through the C ABI; no GPU needed) and against the oracle's independent loader
(oracle/scene_xml.py).  This pins SURVEY row X1 (error behaviour, nesting rules, value parsing, $parameters) to the reference's numbers
instead of read-through."""
import json
import os
import re

import pytest

from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "reference_xml_cases.json")))["records"]


def outcome(load, rec):
    try:
        load(rec["xml"], **rec["kwargs"])
    except Exception as e:     # noqa: BLE001 -- the loaders raise their own exception types
        return "error", str(e)
    return "ok", ""


def check(load, rec):
    got, msg = outcome(load, rec)
    if rec["expect"] == "ok":
        return got == "ok", msg
    if got != "error":
        return False, "loaded without an error"
    return (rec["pattern"] is None or re.search(rec["pattern"], msg) is not None), msg


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_loader_reproduces_the_reference_xml_tests(mi, orc, which):
    load = (lambda xml, **kw: mi.load_string(xml, **kw)) if which == "product" else (lambda xml, **kw: orc.Scene(xml, kw, is_string=True))
    bad = []
    for rec in CASES:
        ok, msg = check(load, rec)
        if not ok:
            bad.append((rec["test"], rec["line"], rec["expect"], rec["pattern"], msg[:160]))
    assert len(CASES) >= 59
    assert not bad, "%d of %d reference XML cases differ:\n%s" % (len(bad), len(CASES), "\n".join(map(str, bad)))
