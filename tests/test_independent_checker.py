"""The numpy checker of tests/independent.py (per-edge Jacobians from the dense restatement + numpy sums, nothing of the
product) against the dense restatement itself at config 1, and against the CPU twin's solve at config 2."""
import numpy as np

from oracle import oracle
from tests import independent, util
from toyslam_amd import synth


def test_checker_reproduces_the_dense_linearisation_of_config_1():
    g = util.c1_arrays()
    d_ref, err, diag_ref, grad_ref = util.dense_solution(g)
    lin = independent.Linearisation(g)
    assert lin.n_tail == 1623                                   # 1623 of 2123 edges in the Huber tail
    assert abs(lin.chi2 - err) <= 1e-12 * err
    np.testing.assert_allclose(lin.gradient(), grad_ref, atol=1e-10 * np.abs(grad_ref).max())
    np.testing.assert_allclose(lin.diag_blocks(), diag_ref, atol=1e-10 * np.abs(diag_ref).max())
    # H @ delta against the dense H of the restatement
    H, b, _, idx = oracle.linearize(util.to_oracle(g))
    x = np.linalg.solve(H, b)
    assert lin.residual_of(d_ref) < 1e-9
    rng = np.random.default_rng(0)
    v = rng.standard_normal((len(g.v_id), 3)); v[g.v_type == 1, 2] = 0
    flat = np.concatenate([v[i, :3 if g.v_type[i] == 0 else 2] for i in range(len(g.v_id))])
    hv = H @ flat
    got = lin.apply_H(v)
    got_flat = np.concatenate([got[i, :3 if g.v_type[i] == 0 else 2] for i in range(len(g.v_id))])
    np.testing.assert_allclose(got_flat, hv, atol=1e-9 * np.abs(hv).max())
    assert np.abs(x - np.concatenate([d_ref[i, :3 if g.v_type[i] == 0 else 2] for i in range(len(g.v_id))])).max() == 0


def test_checker_certifies_the_twin_step_at_config_2():
    g = synth.make_config("c2_10k")
    lin = independent.Linearisation(g)
    ref = oracle.sparse_step(util.to_oracle(g), 1e-12, precond="amg")
    assert abs(lin.chi2 - ref["chi2"]) <= 1e-11 * ref["chi2"]
    assert lin.residual_of(ref["delta"]) < 1e-8
    # and it does tell a wrong step from a right one
    bad = ref["delta"].copy(); bad[5000, 0] += 1e-3 * np.abs(bad).max()
    assert lin.residual_of(bad) > 1e-6
