"""TEST INFRASTRUCTURE: the preparation of a string list for K7 in plain Python -- the three forms as symbol ranks, the
distinct tokens with ids / lengths / tags, the character-class histograms.  It states what k7_fuzz.hip's device kernels
(tokenise, token table, pack) must produce; tests/test_k7_core_cpu.py feeds it to the CPU build of k7_core.h."""
import collections

import numpy as np

N_CLASSES = 32


def forms_of(s):
    toks = s.split()
    distinct = sorted(set(toks))
    return s, " ".join(sorted(toks)), " ".join(distinct), distinct, len(toks)


class Alphabet:
    """symbol ranks 1..n of the to-list's code points (+ the joining space), character classes by frequency"""

    def __init__(self, to_list):
        count = collections.Counter("".join(to_list))
        count.setdefault(" ", 0)
        self.chars = sorted(count)
        self.rank = {c: i + 1 for i, c in enumerate(self.chars)}
        by_freq = sorted(self.chars, key=lambda c: (-count[c], c))
        self.cls = {c: i % N_CLASSES for i, c in enumerate(by_freq)}
        self.n_sym = len(self.chars)
        self.tokens = {}
        for s in to_list:
            for t in sorted(set(s.split())):
                self.tokens.setdefault(t, len(self.tokens))


def prepare(strings, alpha, is_to_list):
    """dict of the arrays tests/k7_core_host.cpp takes for one list"""
    sym = [[], [], []]
    off = [[0], [0], [0]]
    tag, tok_off, tok_id, tok_len, hist, usum = [], [0], [], [], [], []
    unknown = -2
    for s in strings:
        f0, f1, f2, distinct, n_tok_all = forms_of(s)
        for v, f in enumerate((f0, f1, f2)):
            sym[v].extend(alpha.rank.get(c, 0) for c in f)
            off[v].append(len(sym[v]))
        for j, t in enumerate(distinct):
            tag.extend([j & 31] * len(t))
            if j + 1 < len(distinct):
                tag.append((j & 31) | 0x80)
            if t in alpha.tokens:
                tok_id.append(alpha.tokens[t])
            else:
                assert not is_to_list
                tok_id.append(unknown)
                unknown -= 1
            tok_len.append(len(t))
        tok_off.append(len(tok_id))
        h = [0] * N_CLASSES
        for c in f0:
            if c in alpha.cls:
                h[alpha.cls[c]] += 1
        h[alpha.cls[" "]] += max(0, (n_tok_all - 1) - f0.count(" "))     # form 1 may hold more joining spaces than the string
        if max(h) > 255:
            hist.append([0] * (N_CLASSES // 4))
            usum.append(-1)
        else:
            hist.append([h[4 * d] | h[4 * d + 1] << 8 | h[4 * d + 2] << 16 | h[4 * d + 3] << 24 for d in range(N_CLASSES // 4)])
            usum.append(sum(h))
    # symbol presence: 64 bits, bit = symbol rank mod 64, the space left out (see fz_presence_miss; ranks are code-point
    # order: dealing the symbols to the bits by frequency instead was measured WORSE -- a rare symbol that shares its bit
    # with a frequent one is never missed)
    pres = np.zeros((len(strings), 2), np.uint32)
    for i, s in enumerate(strings):
        for c in set(s):
            r = alpha.rank.get(c, 0)
            if r and not c.isspace():
                pres[i, (r & 63) >> 5] |= np.uint32(1 << (r & 31))
    return {"n": len(strings), "pres": pres, "sym": [np.array(x or [0], np.uint16) for x in sym], "off": [np.array(x, np.int64) for x in off],
            "tag": np.array(tag or [0], np.uint8), "tok_off": np.array(tok_off, np.int64), "tok_id": np.array(tok_id or [0], np.int32),
            "tok_len": np.array(tok_len or [0], np.int32), "hist": np.array(hist, np.uint32).reshape(len(strings), N_CLASSES // 4),
            "usum": np.array(usum, np.int32)}
