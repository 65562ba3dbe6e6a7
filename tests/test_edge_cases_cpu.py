"""Edge-case graphs: the sparse CPU twin (both preconditioners) against the dense `cpu eigen` restatement."""
import numpy as np
import pytest

from oracle import oracle
from tests import edge_cases, util


@pytest.mark.parametrize("name", sorted(edge_cases.CASES))
@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_twin_matches_dense_on_edge_cases(name, precond):
    g = edge_cases.CASES[name]()
    ref = oracle.optimize(util.to_oracle(g), 8, mode="cpp", solver="qr")
    r = oracle.sparse_optimize(util.to_oracle(g), 8, pcg_tol=1e-13, precond=precond)
    assert r["iters"] == ref["iters"] and r["stop"] == ref["stop"]
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-9)
    assert util.max_vertex_diff(r["v_pos"], ref["v_pos"], g.v_type) < 1e-8


def test_vertex_and_edge_order_do_not_matter():
    g = edge_cases.base()
    gs, pv = edge_cases.shuffled_vertices_and_edges()
    a = oracle.sparse_optimize(util.to_oracle(g), 6, pcg_tol=1e-13, precond="amg")
    b = oracle.sparse_optimize(util.to_oracle(gs), 6, pcg_tol=1e-13, precond="amg")
    np.testing.assert_allclose(a["chi2"], b["chi2"], rtol=1e-10)
    assert util.max_vertex_diff(a["v_pos"][pv], b["v_pos"], gs.v_type) < 1e-9


@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_gauge_free_graph_keeps_the_chi2_trajectory_of_the_rank_revealing_qr(precond):
    """No fixed vertex: H is singular (a rigid motion of everything costs nothing).  The reference's column-pivoted QR
    returns one of the solutions; PCG on the consistent singular system returns another (a different rigid drift) with
    the SAME chi^2 trajectory — the quantity that does not depend on the gauge.  The multigrid hierarchy's coarsest
    matrix is singular there, the solve breaks down and is repeated with block-Jacobi."""
    g = edge_cases.no_fixed_vertex()
    ref = oracle.optimize(util.to_oracle(g), 6, mode="cpp", solver="qr")
    r = oracle.sparse_optimize(util.to_oracle(g), 6, pcg_tol=1e-12, precond=precond)
    assert r["iters"] == ref["iters"] and r["stop"] == ref["stop"]
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-8)


def test_every_stop_rule_of_the_reference_is_reached():
    """OptimizerCpu.h:140-153 (error rose three times), :167-171 (plateau), :173-177 (short step), the iteration cap:
    the dense restatement and the sparse twin stop in the same iteration for the same reason."""
    cases = {"worse": (edge_cases.pose_graph_without_landmarks(), 50), "plateau": (util.c1_arrays(), 50),
             "converged": (edge_cases.near_optimum(), 50), "cap": (util.c1_arrays(), 7)}
    for want, (g, n) in cases.items():
        ref = oracle.optimize(util.to_oracle(g), n, mode="cpp", solver="chol")
        r = oracle.sparse_optimize(util.to_oracle(g), n, pcg_tol=1e-13, precond="amg")
        assert ref["stop"] == want, (want, ref["stop"])
        assert (r["stop"], r["iters"]) == (ref["stop"], ref["iters"]), (want, r["stop"], r["iters"], ref["iters"])
