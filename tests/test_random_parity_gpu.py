"""Randomised parity sweeps on the GPU (seeded, so a failure reproduces): many small shapes of the
sparse cosine top-n (K3) and random string lists through the device vectoriser (K1/K2), each against
the oracle.  Complements the fixed cases of test_k3_cossim_gpu.py / test_vectorize_gpu.py."""
import numpy as np
import pytest

from tests.helpers import assert_topn_parity, random_csr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_k3_random_shapes(ctx, oracle_mod, seed):
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(1000 + seed)
    n_col = int(rng.integers(3, 700))
    n_a = int(rng.integers(1, 400))
    n_b = int(rng.integers(1, 9000))                 # up to five 2048-row to-blocks
    dens = float(rng.choice([0.01, 0.05, 0.2, 0.6]))
    ntop = int(rng.choice([1, 2, 5, 8, 9, 33, 64, 100]))
    lb = float(rng.choice([0.0, 0.0, 0.2, 0.5, 0.9]))
    empty = set(rng.choice(n_a, size=min(n_a, int(rng.integers(0, 4))), replace=False).tolist())
    a3 = random_csr(rng, n_a, n_col, dens, empty_rows=empty)
    b3 = random_csr(rng, n_b, n_col, dens)
    diag = bool(rng.integers(0, 2)) and n_a <= n_b
    idx, val = _lib.cossim_topn_host(ctx, a3, b3, n_col, ntop, lb, diag)
    exp_idx, exp_val = oracle_mod.cossim_topn(a3, b3, n_col, ntop, lb, exclude_diag=diag)
    assert_topn_parity(idx, val, exp_idx, exp_val, oracle_mod, a3, b3, n_col, exclude_diag=diag,
                       max_near_tie_frac=0.03)
    for r in empty:
        assert (idx[r] == -1).all() and (val[r] == 0).all()


_ALPHABETS = ["abcdefghijklmnopqrstuvwxyz0123456789  ", "abc  ", "AbC dEf-,.;!  ", "añé ü日本語x1  "]


def _random_strings(rng, n, alphabet, max_len):
    chars = np.array(list(alphabet), dtype=object)
    out = []
    for _ in range(n):
        length = int(rng.integers(0, max_len + 1))
        out.append("".join(rng.choice(chars, size=length).tolist()))
    return out


@pytest.mark.parametrize("seed", range(8))
def test_vectoriser_random_strings(ctx, oracle_mod, seed):
    from polyfuzz_amd import _lib
    rng = np.random.default_rng(2000 + seed)
    alphabet = _ALPHABETS[seed % len(_ALPHABETS)]
    clean = bool(seed & 1) and seed % len(_ALPHABETS) != 3   # clean=1 expects 1-byte code units
    lo = int(rng.integers(1, 4))
    hi = int(rng.integers(lo, 5))
    fl = _random_strings(rng, int(rng.integers(1, 1500)), alphabet, int(rng.choice([5, 30, 90])))
    tl = _random_strings(rng, int(rng.integers(1, 2500)), alphabet, int(rng.choice([5, 30, 90])))
    o = oracle_mod.TfidfOracle(n_gram_range=(lo, hi), clean=clean)
    try:
        o.fit(tl + fl)
    except ValueError:            # empty vocabulary: the device must say the same
        with pytest.raises(ValueError, match="empty vocabulary"):
            _lib.DeviceTfidf.fit(ctx, _lib.TfidfParams(lo, hi, int(clean), 1), _lib.DeviceStrings.upload(ctx, tl),
                                 _lib.DeviceStrings.upload(ctx, fl))
        return
    params = _lib.TfidfParams(lo, hi, int(clean), 1)
    f, t = _lib.DeviceStrings.upload(ctx, fl), _lib.DeviceStrings.upload(ctx, tl)
    vec = _lib.DeviceTfidf.fit(ctx, params, t, f)
    _, idf, df = vec.export()
    np.testing.assert_array_equal(df, o.df)
    np.testing.assert_allclose(idf, o.idf, rtol=1e-15)
    for dev, exp in ((vec.transform(f).download(), o.transform(fl)), (vec.transform(t).download(), o.transform(tl))):
        indptr, indices, data, ncols = dev
        assert ncols == len(o.vocabulary)
        np.testing.assert_array_equal(indptr, exp[0])
        np.testing.assert_array_equal(indices, exp[1])
        np.testing.assert_allclose(data, exp[2], rtol=0, atol=2e-7)
