"""Interface parity with the REAL reference package (build-box test: skipped where /root/reference is absent).

The reference's classes are imported in a child process (with the two absent third-party modules stubbed exactly
as tests/golden/make_golden.py does) BEFORE polyfuzz_amd, so that polyfuzz_amd.models.BaseMatcher resolves to the
reference's own ABC, and compared with the mirrors: constructor / match() signatures (names, order, defaults),
the plug-in relation the facade tests with isinstance (polyfuzz/polyfuzz.py:141-152), the function signatures of
cosine_similarity / single_linkage / precision_recall_curve.  No device call is made.
"""
import ast
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "polyfuzz")),
                                reason="the reference package is only present in the build container")

CHILD = r"""
import inspect, json, sys, types
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(ref)r)
sns = types.ModuleType("seaborn"); sys.modules["seaborn"] = sns
rf = types.ModuleType("rapidfuzz"); fuzz = types.ModuleType("rapidfuzz.fuzz"); process = types.ModuleType("rapidfuzz.process")
def ratio(a, b, **kw): return 0.0
def WRatio(a, b, **kw): return 0.0
ratio.__module__ = WRatio.__module__ = "rapidfuzz.fuzz"
fuzz.ratio, fuzz.WRatio = ratio, WRatio
process.extractOne = lambda *a, **k: None
rf.fuzz, rf.process = fuzz, process
sys.modules.update({"rapidfuzz": rf, "rapidfuzz.fuzz": fuzz, "rapidfuzz.process": process})

import polyfuzz                                   # the reference, first
from polyfuzz import PolyFuzz
import polyfuzz.models as ref
from polyfuzz.models._utils import cosine_similarity as ref_cos
from polyfuzz.linkage import single_linkage as ref_link
from polyfuzz.metrics import precision_recall_curve as ref_pr
import polyfuzz_amd.models as ours
from polyfuzz_amd.models import cosine_similarity as our_cos
from polyfuzz_amd.linkage import single_linkage as our_link
from polyfuzz_amd.metrics import precision_recall_curve as our_pr


def sig(f):
    out = []
    for name, p in inspect.signature(f).parameters.items():
        d = p.default
        if d is inspect.Parameter.empty: d = "<required>"
        elif callable(d): d = "callable:" + getattr(d, "__name__", "?")
        out.append([name, str(p.kind), repr(d) if not isinstance(d, str) else d])
    return out

res = {"sig": {}, "rel": {}}
for cls in ("TFIDF", "EditDistance", "RapidFuzz"):
    r, o = getattr(ref, cls), getattr(ours, cls)
    res["sig"][cls + ".__init__"] = [sig(r.__init__), sig(o.__init__)]
    res["sig"][cls + ".match"] = [sig(r.match), sig(o.match)]
    res["rel"][cls + " is a reference BaseMatcher"] = issubclass(o, ref.BaseMatcher)
res["sig"]["Embeddings.match"] = [None, sig(ours.Embeddings.match)]
res["sig"]["Embeddings.__init__"] = [None, sig(ours.Embeddings.__init__)]
res["rel"]["Embeddings is a reference BaseMatcher"] = issubclass(ours.Embeddings, ref.BaseMatcher)
res["rel"]["BaseMatcher is the reference's"] = ours.BaseMatcher is ref.BaseMatcher
res["sig"]["cosine_similarity"] = [sig(ref_cos), sig(our_cos)]
res["sig"]["single_linkage"] = [sig(ref_link), sig(our_link)]
res["sig"]["precision_recall_curve"] = [sig(ref_pr), sig(our_pr)]
# the facade accepts the mirrors as custom models (polyfuzz.py:141-152: isinstance(method, BaseMatcher))
m = ours.TFIDF(n_gram_range=(3, 3), min_similarity=0, top_n=2, model_id="hip")
pf = PolyFuzz(m)
res["rel"]["PolyFuzz(TFIDF) keeps the instance"] = pf.method is m and isinstance(pf.method, ref.BaseMatcher)
res["rel"]["PolyFuzz([...]) keeps the instances"] = all(
    isinstance(x, ref.BaseMatcher) for x in PolyFuzz([ours.TFIDF(model_id="a"), ours.EditDistance(model_id="b"),
                                                       ours.RapidFuzz(model_id="c")]).method)
res["attrs"] = {c: sorted(k for k in dir(getattr(ours, c)()) if not k.startswith("_"))     # (properties included, not evaluated)
                for c in ("TFIDF", "EditDistance", "RapidFuzz")}
res["ref_attrs"] = {c: sorted(k for k in vars(getattr(ref, c)()).keys() if not k.startswith("_"))
                    for c in ("TFIDF", "EditDistance", "RapidFuzz")}
print("RESULT" + json.dumps(res))
"""


@pytest.fixture(scope="module")
def report():
    env = dict(os.environ, PYTHONPATH="")
    p = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO, "ref": REF}], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


def _names(sig):
    return [(n, k) for n, k, _ in sig]


def test_matchers_plug_into_the_reference_facade(report):
    for what, ok in report["rel"].items():
        assert ok, what


@pytest.mark.parametrize("cls", ["TFIDF", "EditDistance", "RapidFuzz"])
def test_constructor_and_match_signatures_equal_the_reference(report, cls):
    ref_init, our_init = report["sig"][cls + ".__init__"]
    assert _names(ref_init) == _names(our_init), (ref_init, our_init)
    for (n, _, d_ref), (_, _, d_our) in zip(ref_init, our_init):
        if n == "scorer":
            # documented deviation: the default is the NAME of the reference's default callable (rapidfuzz is not a
            # dependency of this package); both resolve to the same device scorer
            assert (cls, d_ref, d_our) in (("EditDistance", "callable:ratio", "ratio"),
                                           ("RapidFuzz", "callable:WRatio", "None")), (cls, d_ref, d_our)
        else:
            assert d_ref == d_our, (cls, n, d_ref, d_our)
    ref_match, our_match = report["sig"][cls + ".match"]
    assert _names(ref_match) == _names(our_match), (ref_match, our_match)
    assert [d for _, _, d in ref_match] == [d for _, _, d in our_match]
    # every public attribute the reference's instance carries is there (equal_lists is internal state of its loops)
    missing = set(report["ref_attrs"][cls]) - set(report["attrs"][cls]) - {"equal_lists"}
    assert not missing, missing


@pytest.mark.parametrize("fn", ["cosine_similarity", "single_linkage", "precision_recall_curve"])
def test_function_signatures_equal_the_reference(report, fn):
    ref_sig, our_sig = report["sig"][fn]
    assert _names(ref_sig) == _names(our_sig), (ref_sig, our_sig)
    assert [d for _, _, d in ref_sig] == [d for _, _, d in our_sig], (ref_sig, our_sig)


def test_embeddings_signature_equals_the_reference_source(report):
    """polyfuzz.models.Embeddings needs flair (absent): its signatures are read from the source instead."""
    with open(os.path.join(REF, "polyfuzz", "models", "_embeddings.py")) as f:
        tree = ast.parse(f.read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Embeddings")
    for fn_name in ("__init__", "match"):
        fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == fn_name)
        names = [a.arg for a in fn.args.args]
        defaults = [ast.literal_eval(d) for d in fn.args.defaults]
        ours = report["sig"]["Embeddings." + fn_name][1]
        assert names == [n for n, _, _ in ours], (names, ours)
        our_defaults = [d for _, _, d in ours if d != "<required>"]
        assert [repr(d) if not isinstance(d, str) else d for d in defaults] == our_defaults, (defaults, our_defaults)
