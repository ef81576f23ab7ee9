"""The INTEGRATION.md section 2 flow on the GPU box: what PolyFuzz(custom matcher) calls, in the facade's order --
match -> fit -> transform(re_train=False) -> group -> save -> load -> transform -- through tests/facade_standin.py,
every frame compared with the oracle path (TF-IDF) / the oracle scorers (edit distance)."""
import numpy as np
import pytest

from tests.facade_standin import FacadeStandIn

pytestmark = pytest.mark.gpu


def _expected_tfidf(oracle, from_list, to_list, fit_lists, top_n):
    """(idx, rounded score) the reference path yields for from_list x to_list with a vectoriser fitted on fit_lists"""
    v = oracle.TfidfOracle()
    v.fit(fit_lists)
    a3, b3 = v.transform(from_list), v.transform(to_list)
    idx, val = oracle.cossim_topn(a3, b3, len(v.vocabulary), top_n, 0.0)
    return idx, val


def _check_frame(df, from_list, to_list, idx, val, top_n):
    assert df["From"].tolist() == list(from_list)
    for r in range(top_n):
        tc, sc = ("To", "Similarity") if r == 0 else (f"To_{r + 1}", f"Similarity_{r + 1}")
        exp_sim = np.round(val[:, r], 3)
        got = df[sc].to_numpy()
        near = np.abs(got - np.where(exp_sim < 0.001, 0.0, exp_sim)) <= 0.001 + 1e-12      # fp32 vs float64 at a rounding edge
        assert near.all()
        for i, t in enumerate(df[tc].tolist()):
            if t is not None and abs(val[i, r] - (val[i, r - 1] if r else 2.0)) > 2e-6 and \
                    (r + 1 >= top_n or abs(val[i, r] - val[i, r + 1]) > 2e-6):
                assert t == to_list[idx[i, r]], (i, r, t)


def test_tfidf_through_the_facade_calls(ctx, oracle_mod, tmp_path):
    from polyfuzz_amd import synth
    from polyfuzz_amd.models import TFIDF
    to_list = synth.company_names(1500, seed=11)
    from_list = synth.company_names(700, seed=12)
    new_from = synth.company_names(300, seed=13)
    pf = FacadeStandIn(TFIDF(n_gram_range=(3, 3), min_similarity=0, top_n=2, model_id="hip-tfidf"))
    pf.match(from_list, to_list)
    df = pf.matches["hip-tfidf"]
    idx, val = _expected_tfidf(oracle_mod, from_list, to_list, to_list + from_list, 2)
    _check_frame(df, from_list, to_list, idx, val, 2)

    pf.fit(from_list, to_list)
    out = pf.transform(new_from)["TF-IDF"]                       # vocabulary / idf / index of the fit, resident
    idx2, val2 = _expected_tfidf(oracle_mod, new_from, to_list, to_list + from_list, 2)
    _check_frame(out, new_from, to_list, idx2, val2, 2)

    pf.group(TFIDF(n_gram_range=(3, 3), min_similarity=0.75), link_min_similarity=0.75)
    g = pf.matches["hip-tfidf"]
    assert "Group" in g.columns and len(g) == len(from_list)
    members = {s for c in pf.clusters["hip-tfidf"].values() for s in c}
    assert members <= set(g["To"].dropna()) and set(pf.cluster_mappings["hip-tfidf"]) == members

    path = str(tmp_path / "model.joblib")
    pf.save(path)
    pf2 = FacadeStandIn.load(path)                               # device handles stayed behind; state is re-created lazily
    out2 = pf2.transform(new_from)["TF-IDF"]
    assert out2.equals(out)
    assert pf2.matches["hip-tfidf"].equals(g)


@pytest.mark.parametrize("kind", ["EditDistance", "RapidFuzz"])
def test_edit_matchers_through_the_facade_calls(ctx, oracle_mod, tmp_path, kind):
    from oracle import fuzz_scorers
    from polyfuzz_amd import synth
    from polyfuzz_amd.models import EditDistance, RapidFuzz
    n_to, n_from, n_new = (400, 150, 60) if kind == "EditDistance" else (160, 50, 25)     # (WRatio's oracle is plain Python)
    to_list = synth.company_names(n_to, seed=21)
    from_list = synth.company_names(n_from, seed=22)
    new_from = synth.company_names(n_new, seed=23)
    if kind == "EditDistance":
        m, scorer = EditDistance(normalize=False, model_id="hip-edit"), fuzz_scorers.ratio
    else:
        m, scorer = RapidFuzz(model_id="hip-edit"), fuzz_scorers.WRatio
    pf = FacadeStandIn(m).fit(from_list, to_list)

    def expect(fl):
        idx, score = fuzz_scorers.extract_one_all(fl, to_list, scorer)
        return [to_list[j] for j in idx], np.array(score) / (100.0 if kind == "RapidFuzz" else 1.0)
    to, sim = expect(from_list)
    df = pf.matches["hip-edit"]
    assert df["To"].tolist() == to and np.array_equal(df["Similarity"].to_numpy(), sim)
    out = pf.transform(new_from)["EditDistance"]
    to, sim = expect(new_from)
    assert out["To"].tolist() == to and np.array_equal(out["Similarity"].to_numpy(), sim)
    path = str(tmp_path / "edit.joblib")
    pf.save(path)
    out2 = FacadeStandIn.load(path).transform(new_from)["EditDistance"]
    assert out2.equals(out)
