from ._base import BaseMatcher
from ._utils import cosine_similarity
from ._tfidf import TFIDF
from ._distance import EditDistance
from ._rapidfuzz import RapidFuzz
from ._embeddings import Embeddings

__all__ = ["BaseMatcher", "cosine_similarity", "TFIDF", "EditDistance", "RapidFuzz", "Embeddings"]
