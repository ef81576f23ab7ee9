"""A stand-in for the reference's PolyFuzz facade (test infrastructure).

The facade itself is out of scope and absent on the GPU box; this class replays, in as few lines
as possible, the CALLS it makes on a custom matcher, so that the INTEGRATION.md section 2 flow can run where the GPU
is: match (polyfuzz/polyfuzz.py:141-152), fit / transform with re_train=False (:196-243), group through
single_linkage on a self-match of the unique To strings (:331-373, 459-484), save / load through joblib (:429-457).
"""
import joblib

from polyfuzz_amd.linkage import single_linkage
from polyfuzz_amd.models import BaseMatcher


class FacadeStandIn:
    def __init__(self, method):
        assert isinstance(method, BaseMatcher)          # the facade's own test for a custom model
        self.method, self.matches, self.to_list = method, None, None
        self.clusters, self.cluster_mappings = {}, {}

    def match(self, from_list, to_list=None):
        self.matches = {self.method.model_id: self.method.match(from_list, to_list)}
        return self

    def fit(self, from_list, to_list=None):
        self.match(from_list, to_list)
        self.to_list = to_list if to_list is not None else from_list
        return self

    def transform(self, from_list):
        return {self.method.type: self.method.match(from_list, self.to_list, re_train=False)}

    def group(self, model, link_min_similarity=0.75, group_all_strings=False):
        for name, df in self.matches.items():
            col = df.From if group_all_strings else df.To
            strings = list(col.dropna().unique())
            clusters, id_map, name_map = single_linkage(model.match(strings), link_min_similarity)
            df["Group"] = df["To"].map(name_map).fillna(df["To"])
            self.matches[name], self.clusters[name], self.cluster_mappings[name] = df, clusters, id_map
        return self

    def save(self, path):
        with open(path, "wb") as f:
            joblib.dump(self, f)

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return joblib.load(f)
