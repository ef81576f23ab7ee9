"""
oracle/fuzz_scorers.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-Python restatement of the rapidfuzz.fuzz scorers the reference's RapidFuzz matcher can be given
(polyfuzz/models/_rapidfuzz.py:45-58, 106-108; default fuzz.WRatio): ratio, partial_ratio, token_sort_ratio,
token_set_ratio, token_ratio, partial_token_sort_ratio, partial_token_set_ratio, partial_token_ratio, WRatio, QRatio,
with `process.extractOne`'s rule on top (first best choice).

PARITY UNPINNED: rapidfuzz (setup.py:20, `rapidfuzz>=0.13.1`, un-vendored, not installable here) holds the algorithm;
this file restates the published semantics of rapidfuzz 3.x (its pure-Python fallback `fuzz_py.py` states them most
plainly): no default processor, whitespace tokens, `" ".join(sorted(...))`, the three-part partial_ratio window sweep,
the 0.95 / 0.9 / 0.6 WRatio scales.  Anchored on the values rapidfuzz publishes in its README / API docs
(tests/test_fuzz_oracle_cpu.py).  Two normalisations appear in rapidfuzz and are kept apart here because they round
differently in float64:  ratio-like scores are (1 - dist / lensum) * 100,  token_set's are 100 - 100 * dist / lensum.

Plain Python: small cases only (the LCS itself is a big-integer bit-vector, pinned on the O(|a||b|) DP).
"""
from typing import Callable, List, Optional, Sequence, Tuple


def lcs_len_dp(a: Sequence, b: Sequence) -> int:
    """The definition: plain O(|a||b|) dynamic programming."""
    if not a or not b:
        return 0
    prev = [0] * (len(b) + 1)
    for ca in a:
        cur = [0]
        for j, cb in enumerate(b):
            cur.append(prev[j] + 1 if ca == cb else max(prev[j + 1], cur[j]))
        prev = cur
    return prev[-1]


def lcs_len(a: Sequence, b: Sequence) -> int:
    """The same number from one big-integer bit-vector (Hyyro's recurrence) -- partial_ratio calls this once per
    window and the DP above makes the test suite crawl.  tests/test_fuzz_oracle_cpu.py holds the two equal on
    thousands of random pairs."""
    if not a or not b:
        return 0
    pm = {}
    for i, c in enumerate(a):
        pm[c] = pm.get(c, 0) | (1 << i)
    full = (1 << len(a)) - 1
    v = full
    for c in b:
        u = v & pm.get(c, 0)
        v = ((v + u) | (v - u)) & full
    return len(a) - bin(v).count("1")


def indel_distance(a: Sequence, b: Sequence) -> int:
    return len(a) + len(b) - 2 * lcs_len(a, b)


def _ratio_of(dist: int, lensum: int) -> float:
    """Indel.normalized_similarity * 100 (rapidfuzz: norm_dist = dist / maximum, 0 when maximum == 0)"""
    norm_dist = dist / lensum if lensum else 0.0
    return (1.0 - norm_dist) * 100


def _norm_distance(dist: int, lensum: int) -> float:
    """rapidfuzz's norm_distance<100> (token_set_ratio)"""
    return 100 - 100 * dist / lensum if lensum else 100.0


def ratio(s1: str, s2: str) -> float:
    return _ratio_of(indel_distance(s1, s2), len(s1) + len(s2))


def QRatio(s1: str, s2: str) -> float:
    if not s1 or not s2:
        return 0.0
    return ratio(s1, s2)


def _partial_ratio_impl(s1: str, s2: str) -> float:
    """len(s1) <= len(s2): the best ratio of s1 against the prefixes of s2 shorter than s1, its windows of length
    len(s1) and its suffixes shorter than s1 (fuzz_py._partial_ratio_impl; its character-set test only skips windows
    that another window dominates)."""
    len1, len2 = len(s1), len(s2)
    best = 0.0
    for i in range(1, len1):
        best = max(best, ratio(s1, s2[:i]))
    for i in range(len2 - len1):
        best = max(best, ratio(s1, s2[i:i + len1]))
    for i in range(len2 - len1, len2):
        best = max(best, ratio(s1, s2[i:]))
    return best


def partial_ratio(s1: str, s2: str) -> float:
    if not s1 or not s2:            # (rapidfuzz: 100 when both are empty, else 0 -- no window of "" is scored)
        return 100.0 if not s1 and not s2 else 0.0
    shorter, longer = (s1, s2) if len(s1) <= len(s2) else (s2, s1)
    res = _partial_ratio_impl(shorter, longer)
    if res != 100 and len(s1) == len(s2):
        res = max(res, _partial_ratio_impl(longer, shorter))
    return res


def _sorted_join(tokens) -> str:
    return " ".join(sorted(tokens))


def token_sort_ratio(s1: str, s2: str) -> float:
    return ratio(_sorted_join(s1.split()), _sorted_join(s2.split()))


def token_set_ratio(s1: str, s2: str) -> float:
    tokens_a, tokens_b = set(s1.split()), set(s2.split())
    if not tokens_a or not tokens_b:
        return 0.0
    intersect = tokens_a & tokens_b
    diff_ab, diff_ba = tokens_a - tokens_b, tokens_b - tokens_a
    if intersect and (not diff_ab or not diff_ba):
        return 100.0
    diff_ab_joined, diff_ba_joined = _sorted_join(diff_ab), _sorted_join(diff_ba)
    ab_len, ba_len = len(diff_ab_joined), len(diff_ba_joined)
    sect_len = len(_sorted_join(intersect))
    sect_ab_len = sect_len + (sect_len != 0) + ab_len
    sect_ba_len = sect_len + (sect_len != 0) + ba_len
    result = _norm_distance(indel_distance(diff_ab_joined, diff_ba_joined), sect_ab_len + sect_ba_len)
    if not sect_len:
        return result
    sect_ab_ratio = _norm_distance((sect_len != 0) + ab_len, sect_len + sect_ab_len)
    sect_ba_ratio = _norm_distance((sect_len != 0) + ba_len, sect_len + sect_ba_len)
    return max(result, sect_ab_ratio, sect_ba_ratio)


def token_ratio(s1: str, s2: str) -> float:
    return max(token_sort_ratio(s1, s2), token_set_ratio(s1, s2))


def partial_token_sort_ratio(s1: str, s2: str) -> float:
    return partial_ratio(_sorted_join(s1.split()), _sorted_join(s2.split()))


def partial_token_set_ratio(s1: str, s2: str) -> float:
    tokens_a, tokens_b = set(s1.split()), set(s2.split())
    if not tokens_a or not tokens_b:
        return 0.0
    if tokens_a & tokens_b:
        return 100.0
    return partial_ratio(_sorted_join(tokens_a - tokens_b), _sorted_join(tokens_b - tokens_a))


def partial_token_ratio(s1: str, s2: str) -> float:
    tokens_a, tokens_b = set(s1.split()), set(s2.split())
    if not tokens_a or not tokens_b:
        return 0.0
    if tokens_a & tokens_b:
        return 100.0
    return max(partial_token_sort_ratio(s1, s2), partial_token_set_ratio(s1, s2))


def WRatio(s1: str, s2: str) -> float:
    UNBASE_SCALE = 0.95
    if not s1 or not s2:
        return 0.0
    len1, len2 = len(s1), len(s2)
    len_ratio = len1 / len2 if len1 > len2 else len2 / len1
    end_ratio = ratio(s1, s2)
    if len_ratio < 1.5:
        return max(end_ratio, token_ratio(s1, s2) * UNBASE_SCALE)
    PARTIAL_SCALE = 0.9 if len_ratio < 8.0 else 0.6
    end_ratio = max(end_ratio, partial_ratio(s1, s2) * PARTIAL_SCALE)
    return max(end_ratio, partial_token_ratio(s1, s2) * UNBASE_SCALE * PARTIAL_SCALE)


SCORERS = {f.__name__: f for f in (ratio, QRatio, partial_ratio, token_sort_ratio, token_set_ratio, token_ratio,
                                   partial_token_sort_ratio, partial_token_set_ratio, partial_token_ratio, WRatio)}


def extract_one_all(from_list: List[str], to_list: List[str], scorer: Callable[[str, str], float],
                    skip: Optional[Sequence[int]] = None) -> Tuple[List[int], List[float]]:
    """process.extractOne for every from-string: the FIRST choice with the highest score; `skip[i]` is a choice index
    left out for from-string i (a self-match's own first occurrence).  -1 / 0.0 when there is no choice."""
    idx, score = [], []
    for i, s in enumerate(from_list):
        best_j, best = -1, -1.0
        for j, t in enumerate(to_list):
            if skip is not None and j == skip[i]:
                continue
            v = scorer(s, t)
            if v > best:
                best_j, best = j, v
        idx.append(best_j)
        score.append(best if best_j >= 0 else 0.0)
    return idx, score
