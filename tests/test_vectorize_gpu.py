"""GPU parity of K1/K2 (device TF-IDF vectorisation) against the oracle restatement of
sklearn's TfidfVectorizer + the reference analyzer (oracle/tfidf_oracle.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

README_FROM = ["apple", "apples", "appl", "recal", "house", "similarity"]
README_TO = ["apple", "apples", "mouse"]


def _device_vectorize(ctx, from_list, to_list, lo, hi, clean, remove_space=True):
    from polyfuzz_amd import _lib
    params = _lib.TfidfParams(lo, hi, int(clean), int(remove_space))
    f = _lib.DeviceStrings.upload(ctx, from_list)
    t = _lib.DeviceStrings.upload(ctx, to_list) if to_list is not None else None
    vec = _lib.DeviceTfidf.fit(ctx, params, t if t is not None else f, f if t is not None else None)
    out = [vec.transform(f).download()]
    if t is not None:
        out.append(vec.transform(t).download())
    return vec, out


def _check_csr(dev, exp, n_col):
    indptr, indices, data, ncols = dev
    e_indptr, e_indices, e_data = exp
    assert ncols == n_col
    np.testing.assert_array_equal(indptr, e_indptr)
    np.testing.assert_array_equal(indices, e_indices)
    np.testing.assert_allclose(data, e_data, rtol=0, atol=2e-7)
    return float((data == e_data.astype(np.float32)).mean()) if len(data) else 1.0


@pytest.mark.parametrize("rng", [(1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (3, 3), (3, 5)])
@pytest.mark.parametrize("clean", [True, False])
def test_readme_lists_all_ranges(ctx, oracle_mod, rng, clean):
    vec, (a, b) = _device_vectorize(ctx, README_FROM, README_TO, rng[0], rng[1], clean)
    o = oracle_mod.TfidfOracle(n_gram_range=rng, clean=clean).fit(README_TO + README_FROM)
    _check_csr(a, o.transform(README_FROM), len(o.vocabulary))
    _check_csr(b, o.transform(README_TO), len(o.vocabulary))
    ngrams, idf, df = vec.export()
    names = ["".join(chr(c) for c in row if c) for row in ngrams.tolist()]
    assert names == o.vocabulary
    np.testing.assert_array_equal(df, o.df)
    np.testing.assert_allclose(idf, o.idf, rtol=1e-15)


def test_company_c2(ctx, oracle_mod, golden):
    fl = golden["company_c2_lists"]["from_list"]
    tl = golden["company_c2_lists"]["to_list"]
    vec, (a, b) = _device_vectorize(ctx, fl, tl, 3, 3, True)
    o = oracle_mod.TfidfOracle().fit(tl + fl)
    assert len(o.vocabulary) == int(golden["npz"]["c2_vocab_size"][0])
    exact_a = _check_csr(a, o.transform(fl), len(o.vocabulary))
    exact_b = _check_csr(b, o.transform(tl), len(o.vocabulary))
    assert a[0][-1] == golden["npz"]["c2_nnz"][0] and b[0][-1] == golden["npz"]["c2_nnz"][1]
    assert min(exact_a, exact_b) > 0.999   # fp32(float64 value) bit-for-bit on (nearly) every entry


def test_document_frequencies_both_counting_paths(ctx, oracle_mod, golden, monkeypatch):
    """df over 20k strings (several LDS-histogram workgroups) equals the oracle's, and the global-atomic
    kernels kept for vocabularies beyond the LDS histogram (forced by PFZ_NO_LDS_HIST) give the same
    vectoriser and the same matrices bit for bit."""
    fl = golden["company_c2_lists"]["from_list"]
    tl = golden["company_c2_lists"]["to_list"]
    o = oracle_mod.TfidfOracle().fit(tl + fl)
    vec, (a, b) = _device_vectorize(ctx, fl, tl, 3, 3, True)
    _, idf, df = vec.export()
    np.testing.assert_array_equal(df, o.df)
    np.testing.assert_allclose(idf, o.idf, rtol=1e-15)
    monkeypatch.setenv("PFZ_NO_LDS_HIST", "1")
    vec2, (a2, b2) = _device_vectorize(ctx, fl, tl, 3, 3, True)
    _, idf2, df2 = vec2.export()
    np.testing.assert_array_equal(df2, df)
    np.testing.assert_array_equal(idf2, idf)
    for x, y in zip(a + b, a2 + b2):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("rng,clean", [((3, 7), True), ((6, 10), True), ((2, 6), False), ((4, 4), False)])
def test_wide_ngram_codes_use_the_sorted_vocabulary(ctx, oracle_mod, golden, rng, clean):
    """n-gram codes wider than the presence bitmap addresses (> 32 bits: long n-grams, big alphabets) go
    through the sorted-vocabulary path (radix sort + unique, binary-search ranks).  Same contract: the
    oracle's vocabulary order, df, idf and matrices; export -> import round trip."""
    from polyfuzz_amd import _lib
    fl = golden["company_c2_lists"]["from_list"][:1500]
    tl = golden["company_c2_lists"]["to_list"][:2000]
    if not clean:       # a large alphabet: 6 bits x 6 = 36 bits, and with the wide characters 9+ bits x 4
        t = golden["titles_lists"]
        fl = fl[:700] + t["from_list"][:700] + ["\u65e5\u672c\u8a9e\u30c6\u30b9\u30c8 \u6771\u4eac", "\u00fcber stra\u00dfe"]
        tl = tl[:900] + t["to_list"][:900] + ["\u6771\u4eac\u30c6\u30b9\u30c8", "stra\u00dfe"]
        extra = "".join(chr(0x400 + i) for i in range(600))          # pad the alphabet past 511 symbols
        tl = tl + [extra[i:i + 40] for i in range(0, 600, 40)]
    vec, (a, b) = _device_vectorize(ctx, fl, tl, rng[0], rng[1], clean)
    assert vec.info()["code_bits"] > 32
    o = oracle_mod.TfidfOracle(n_gram_range=rng, clean=clean).fit(tl + fl)
    _check_csr(a, o.transform(fl), len(o.vocabulary))
    _check_csr(b, o.transform(tl), len(o.vocabulary))
    ngrams, idf, df = vec.export()
    names = ["".join(chr(c) for c in row if c) for row in ngrams.tolist()]
    assert names == o.vocabulary
    np.testing.assert_array_equal(df, o.df)
    np.testing.assert_allclose(idf, o.idf, rtol=1e-15)
    params = _lib.TfidfParams(rng[0], rng[1], int(clean), 1)
    vec2 = _lib.DeviceTfidf.from_state(ctx, params, ngrams, idf, vec.info()["n_docs"])
    got2 = vec2.transform(_lib.DeviceStrings.upload(ctx, fl)).download()
    for x, y in zip(a, got2):
        np.testing.assert_array_equal(x, y)


def test_self_fit_and_messy_strings(ctx, oracle_mod):
    docs = ["  Hello,   World!! ", "A\tB  C\nD", "", "   ", "a", "ab", "abc", "ABC abc AbC", "x" * 70 + " " + "yz" * 40,
            "1st & 2nd St.", "trailing   ", "Ünited été", "a  b", "..."]
    for clean in (True, False):
        for rs in (True, False):
            vec, (a,) = _device_vectorize(ctx, docs, None, 2, 3, clean, rs)
            o = oracle_mod.TfidfOracle(n_gram_range=(2, 3), clean=clean, remove_space_ngrams=rs).fit(docs)
            _check_csr(a, o.transform(docs), len(o.vocabulary))


def test_wide_characters_and_long_rows(ctx, oracle_mod, golden):
    t = golden["titles_lists"]
    fl, tl = t["from_list"], t["to_list"]
    long_docs = ["".join(chr(97 + (i * 7 + j) % 26) for j in range(300 + 40 * i)) for i in range(6)]
    long_docs.append("ab" * 3000)            # > 4096 n-grams: the global-scratch path of k_rows_long
    fl = fl + long_docs
    vec, (a, b) = _device_vectorize(ctx, fl, tl, 3, 3, False)
    o = oracle_mod.TfidfOracle(clean=False).fit(tl + fl)
    _check_csr(a, o.transform(fl), len(o.vocabulary))
    _check_csr(b, o.transform(tl), len(o.vocabulary))


def test_transform_out_of_vocabulary_and_roundtrip(ctx, oracle_mod):
    from polyfuzz_amd import _lib
    fit_docs = ["apple pie", "apple tart", "cherry pie"]
    new_docs = ["apple strudel", "zzz qqq", "", "pie"]
    for clean in (True, False):
        params = _lib.TfidfParams(3, 3, int(clean), 1)
        vec = _lib.DeviceTfidf.fit(ctx, params, _lib.DeviceStrings.upload(ctx, fit_docs), None)
        o = oracle_mod.TfidfOracle(clean=clean).fit(fit_docs)
        got = vec.transform(_lib.DeviceStrings.upload(ctx, new_docs)).download()
        _check_csr(got, o.transform(new_docs), len(o.vocabulary))
        ngrams, idf, df = vec.export()
        vec2 = _lib.DeviceTfidf.from_state(ctx, params, ngrams, idf, vec.info()["n_docs"])
        got2 = vec2.transform(_lib.DeviceStrings.upload(ctx, new_docs)).download()
        for x, y in zip(got, got2):
            np.testing.assert_array_equal(x, y)


def test_errors(ctx):
    from polyfuzz_amd import _lib
    s = _lib.DeviceStrings.upload(ctx, ["", "  ", "ab"])
    with pytest.raises(ValueError, match="empty vocabulary"):
        _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 3, 1, 1), s, None)
    with pytest.raises(NotImplementedError):
        _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 11, 1, 1), s, None)     # 66-bit codes: more than a uint64
    with pytest.raises(_lib.PfzError):
        _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(3, 2, 1, 1), s, None)


@pytest.mark.parametrize("clean", [True, False])
@pytest.mark.parametrize("rng,remove_space", [((3, 3), True), ((2, 4), True), ((1, 2), False), ((3, 6), True)])
def test_wave_per_string_extract_equals_thread_per_string(ctx, oracle_mod, monkeypatch, clean, rng, remove_space):
    """k_extract_wave (a wave per string, a lane per character: round 4) against k_extract (a thread per string) and the
    oracle, on strings built to sit on its seams: lengths 0 / 1 / 63 / 64 / 65 / 127 / 128 / 129 / 200 / 256 (the chunk
    boundaries of the 64-lane passes), runs of blanks across a boundary, leading / trailing blanks, punctuation and tabs
    that vanish, upper case, digits; more strings than one workgroup takes (128)."""
    r = np.random.default_rng(9)
    alphabet = list("abcdeXYZ0189") + [" "] * 4 + list(",.-\t&'")
    strings = []
    for n in [0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 256] * 6 + list(r.integers(0, 120, 400)):
        strings.append("".join(r.choice(alphabet, size=int(n))))
    strings += [" " * 70 + "ab cd" + " " * 70, "a" + " " * 63 + "b", "ab" + " " * 62 + "cd" + " " * 64 + "ef", "x" * 256, " " * 256]
    out = {}
    for which in ("wave", "thread"):
        monkeypatch.setenv("PFZ_K1_EXTRACT", which)
        vec, (a,) = _device_vectorize(ctx, strings, None, rng[0], rng[1], clean, remove_space)
        out[which] = a
    for x, y in zip(out["wave"][:3], out["thread"][:3]):
        np.testing.assert_array_equal(x, y)                         # the very same CSR, bit for bit
    o = oracle_mod.TfidfOracle(n_gram_range=rng, clean=clean, remove_space_ngrams=remove_space).fit(strings)
    _check_csr(out["wave"], o.transform(strings), len(o.vocabulary))


@pytest.mark.parametrize("clean", [True, False])
def test_rows_of_65_to_128_ngrams(ctx, oracle_mod, clean):
    """Strings of 65 .. 128 n-grams are sorted by one wave with two keys per lane (k_rows_short) instead of a workgroup with a
    barrier per stage (k_rows_long, from 129 on): every count around the seams -- 63 .. 66 and 126 .. 131 n-grams --, strings
    made of few distinct n-grams (long runs of equal keys across the two registers), strings with characters outside the
    fitted vocabulary, 3-grams alone and a range of n; CSR against the oracle, bit for bit in structure."""
    rng = np.random.default_rng(9)
    docs = []
    for n_grams in list(range(61, 70)) + list(range(120, 134)) + [90, 100, 110]:
        length = n_grams + 2                                   # 3-grams of a string without blanks: len - 2
        docs.append("".join(chr(97 + int(c)) for c in rng.integers(0, 26, length)))
        docs.append("".join("ab"[int(c)] for c in rng.integers(0, 2, length)))        # eight distinct 3-grams at most
        docs.append(("xyz" * 60)[:length])                                              # three distinct 3-grams
    docs += ["short", "", "abc"]
    vec, (a,) = _device_vectorize(ctx, docs, None, 3, 3, clean)
    o = oracle_mod.TfidfOracle(n_gram_range=(3, 3), clean=clean).fit(docs)
    _check_csr(a, o.transform(docs), len(o.vocabulary))
    # out-of-vocabulary n-grams at transform time, and two n values (R = 2 slots per character)
    fit_docs = docs[::2]
    vec2, _ = _device_vectorize(ctx, fit_docs, None, 2, 3, clean)
    from polyfuzz_amd import _lib
    new = [d[:40] + "QQ" + d[40:] for d in docs if 45 <= len(d) <= 66]
    got = vec2.transform(_lib.DeviceStrings.upload(ctx, new)).download()
    o2 = oracle_mod.TfidfOracle(n_gram_range=(2, 3), clean=clean).fit(fit_docs)
    _check_csr(got, o2.transform(new), len(o2.vocabulary))


@pytest.mark.parametrize("n", [1, 15, 16, 17, 255, 4097, 16384, 16385])
def test_fused_launches_of_a_transform(ctx, oracle_mod, monkeypatch, n):
    """A transform's launches, fused where the list is short: (i) k_extract_wave<.., ROWS> sorts and counts a string's n-grams
    right behind their extraction (PFZ_K1_FUSE_ROWS=0: the k_rows_short launch); (ii) lists of up to 16 384 strings:
    k_finalize<true> adds up the row counts itself (PFZ_K2_SELF_SCAN=0: copy + scan + finalize).  Same CSR bit for bit whichever
    way, the count of non-zeros (the host's pinned word) included; empty rows at either end and in between, strings of 65 .. 128
    and of more than 128 n-grams (k_rows_long's) in the list; the oracle on the short lists."""
    from polyfuzz_amd import _lib, datasets
    names = datasets.load_company_names()
    if n > 8:
        docs = ["", "!!", " ".join(names[:5]), " ".join(names[5:30])] + names[:n - 7] + ["", "zz", ""]
    else:
        docs = (names[:1] + ["", "ab", "", "x y z"])[:n]
    assert len(docs) == n
    params = _lib.TfidfParams(3, 3, 1, 1)
    vec = _lib.DeviceTfidf.fit(ctx, params, _lib.DeviceStrings.upload(ctx, names[:30000]), None)
    got = {}
    for fuse in ("1", "0"):
        for scan in ("1", "0"):
            monkeypatch.setenv("PFZ_K1_FUSE_ROWS", fuse)
            monkeypatch.setenv("PFZ_K2_SELF_SCAN", scan)
            m = vec.transform(_lib.DeviceStrings.upload(ctx, docs))
            got[fuse + scan] = (m.shape, m.download())
    for key in ("10", "01", "00"):
        assert got[key][0] == got["11"][0], key
        for x, y in zip(got[key][1], got["11"][1]):
            np.testing.assert_array_equal(x, y, err_msg=key)
    indptr = got["11"][1][0]
    assert len(indptr) == n + 1 and got["11"][0][2] == indptr[-1]
    if n <= 4097:
        o = oracle_mod.TfidfOracle().fit(names[:30000])
        _check_csr(got["11"][1], o.transform(docs), len(o.vocabulary))


def test_two_list_fit_in_shared_launches_equals_one_launch_per_list(ctx, oracle_mod, monkeypatch):
    """a fit over to + from extracts both lists, sorts both lists' short rows and counts both lists' document frequencies in ONE
    launch each (PFZ_K1_TWO_LISTS=0: a launch per list): same vocabulary, document frequencies, idf and matrices bit for bit --
    lists of changing sizes (1 .. 12 000), with strings of more than 128 n-grams in either list (k_rows_long's), a to-list too
    long for the wave kernel (40 000 strings: the lists fall back to their own launches), a from-list of wide characters (another
    kernel instance: no sharing); the oracle on two of them."""
    from polyfuzz_amd import _lib, datasets
    names = datasets.load_company_names()
    long_a, long_b = " ".join(names[:25]), " ".join(names[100:140])
    rng = np.random.default_rng(5)
    cases = []
    for trial in range(6):
        n_to, n_from = int(rng.integers(1, 12000)), int(rng.integers(1, 12000))
        lo = int(rng.integers(0, 40000))
        to, frm = list(names[lo:lo + n_to]), list(names[lo + 20000:lo + 20000 + n_from])
        if trial % 3 == 1:
            to.insert(len(to) // 2, long_a)
        if trial % 3 == 2:
            frm.append(long_b)
            frm.insert(0, "")
        cases.append((to, frm, True))
    cases.append((names[:40000], names[50000:53000], True))
    cases.append((names[:3000], [s + " ŝĝ" for s in names[4000:6000]], False))
    for ci, (to, frm, clean) in enumerate(cases):
        params = _lib.TfidfParams(3, 3, int(clean), 1)
        got = {}
        for knob in ("1", "0"):
            monkeypatch.setenv("PFZ_K1_TWO_LISTS", knob)
            t, f = _lib.DeviceStrings.upload(ctx, to), _lib.DeviceStrings.upload(ctx, frm)
            vec = _lib.DeviceTfidf.fit(ctx, params, t, f)
            got[knob] = (vec.export(), vec.transform(f).download(), vec.transform(t).download())
        for x, y in zip(got["1"][0], got["0"][0]):
            np.testing.assert_array_equal(x, y, err_msg=f"case {ci}")
        for m in (1, 2):
            for x, y in zip(got["1"][m], got["0"][m]):
                np.testing.assert_array_equal(x, y, err_msg=f"case {ci}")
        if ci in (1, 2):
            o = oracle_mod.TfidfOracle().fit(list(to) + list(frm))
            _check_csr(got["1"][1], o.transform(frm), len(o.vocabulary))
            _check_csr(got["1"][2], o.transform(to), len(o.vocabulary))
