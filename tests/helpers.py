"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import glob
import json
import os

import numpy as np

from reactranker_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def model_case_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "model_*.npz")))


def load_case(path):
    d = np.load(path, allow_pickle=False)
    cfg = json.loads(str(d["cfg"]))
    return d, cfg


def kmix_queries(seed, scope):
    """Same construction as tools/make_golden.py:kmix_queries (K_r = 2, K_p = 3)."""
    rng = np.random.default_rng(seed)
    r_specs, p_specs, targets = [], [], []
    for c in scope:
        n = int(rng.integers(5, 9))
        r = synth.random_reactant(rng, n, max_degree=2)
        for _ in range(c):
            r_specs.append(r)
            p_specs.append(synth.random_product(rng, r, max_degree=4))
        targets.append(rng.standard_normal(c).astype(np.float32))
    m = sum(scope)
    return synth.QueryBatch(r_specs, p_specs, list(scope), np.concatenate(targets),
                            rng.random((m, 1)).astype(np.float32))


def case_queries(cfg):
    if cfg.get("kmix"):
        return kmix_queries(cfg["seed"], cfg["scope"])
    return synth.make_queries(cfg["seed"], len(cfg["scope"]), cfg["scope"], atoms_lo=5, atoms_hi=12)


def case_weights(d, cfg, shapes):
    if cfg["hidden_size"] >= 300:
        return synth.seeded_weights(shapes, cfg["wseed"])
    return {k[2:]: d[k] for k in d.files if k.startswith("w.")}


def golden_graph(d, prefix):
    return {k: d[prefix + k] for k in ("f_atoms", "f_bonds", "a2b", "b2a", "b2revb", "a2a", "a_scope")}


def sample_like(a, hidden):
    """Golden grads of H>=300 cases are stored as a [::7, ::5] sample of 2-D tensors."""
    if hidden >= 300 and a.ndim == 2:
        return a[::7, ::5]
    return a


# ---------------------------------------------------------------------------------------------- measured-error recorder
# tests/conftest.py points CURRENT at the running test's table; the close() / _compare() helpers of the -m gpu tests feed
# it, so the parity log (profiles/rNN_parity_errors.txt) holds the worst error every parity test MEASURED, next to its bound.
CURRENT = None


def record(what, err, tol=None):
    """Remember the largest error seen under the label `what` in the running test."""
    if CURRENT is None:
        return
    what = str(what) if what else "value"
    e = float(err)
    cur = CURRENT.get(what)
    if cur is None or not (e <= cur[0]):
        CURRENT[what] = (e, tol)
