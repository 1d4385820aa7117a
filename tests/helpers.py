"""Shared test helpers (CPU side)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORE_TOL = 1e-5  # BASELINE.json north_star: cosine scores within 1e-5 (fp32)


def load_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "rank_golden.npz"))


def assert_topk_matches(v, i, v_ref, i_ref, gap=None, cert_gap=1e-5, what="", scores_ref=None):
    """Scores within SCORE_TOL everywhere; indices bit-exact on every row whose reference gaps are
    certified (> cert_gap, far above fp32 summation noise ~1e-7).  On uncertified rows an index may
    differ only where the reference scores of the two candidates are closer than cert_gap; with ``scores_ref``
    (the full reference score matrix) that is checked directly: the returned index's reference score must equal
    the reference's score at that rank within cert_gap (several near-ties in one row are then fine)."""
    v, i, v_ref, i_ref = map(np.asarray, (v, i, v_ref, i_ref))
    assert v.shape == v_ref.shape and i.shape == i_ref.shape, (what, v.shape, v_ref.shape)
    np.testing.assert_allclose(v, v_ref, rtol=0, atol=SCORE_TOL, err_msg=f"{what}: scores")
    rows = np.arange(v.shape[0])
    cert = np.ones(v.shape[0], bool) if gap is None else (np.asarray(gap) > cert_gap)
    bad = rows[cert][(i[cert] != i_ref[cert]).any(1)]
    assert bad.size == 0, f"{what}: index mismatch on certified rows {bad[:8]}"
    for r in rows[~cert]:
        diff = np.nonzero(i[r] != i_ref[r])[0]
        for p in diff:
            assert abs(float(v[r, p]) - float(v_ref[r, p])) <= cert_gap, (what, r, p)
        if scores_ref is not None:
            for p in diff:
                assert abs(float(scores_ref[r][i[r, p]]) - float(v_ref[r, p])) <= cert_gap, (what, r, p)
            continue
        # same multiset up to the near-tied boundary element
        assert len(set(i[r]) ^ set(i_ref[r])) <= 2, (what, r)
    return int(cert.sum())


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mi355_retrieval.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", txt)))
