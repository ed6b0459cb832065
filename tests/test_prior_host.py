"""Host-side prior propagation (vinsat_amd/prior.py) against outputs of the reference's own
``propagate_dynamics_cov_init`` (BA_utils.py:227-248; fixture made by tools/gen_golden.py PRIORPROP)."""
import numpy as np

from conftest import load_golden, rel_err
from vinsat_amd import prior


def test_propagate_dynamics_cov_init_matches_reference():
    g = load_golden("prior_prop")
    st, v, Hs, Hr = prior.propagate_dynamics_cov_init(g["state"], g["velocity"], g["hessian"], g["omega"], int(g["tdiff"]),
                                                      int(g["duration"]), 1)
    assert st.shape == (41, 10) and Hs.shape == (41, 6, 6) and Hr.shape == (41, 3, 3)
    assert rel_err(st, g["states_t"][0]) < 1e-13
    assert rel_err(v, g["velocities_t"][0]) < 1e-13
    assert rel_err(Hs, g["hessian_state_t"][0]) < 1e-11     # two inversions of a matrix of condition ~1e4
    assert rel_err(Hr, g["hessian_rot_t"][0]) < 1e-12


def test_rk4_step_jacobian_is_the_derivative_of_the_step():
    """The closed-form per-step Jacobian against central differences of the step itself."""
    from vinsat_amd.synth import rk4_step
    x = np.array([-6800.0, 300.0, 1200.0, 0.4, -1.2, 7.4])
    _, J = prior.rk4_step_with_jacobian(x)
    num = np.zeros((6, 6))
    for b in range(6):
        h = 1e-2 if b < 3 else 1e-4
        e = np.zeros(6)
        e[b] = h
        num[:, b] = (rk4_step(x + e) - rk4_step(x - e)) / (2 * h)
    assert np.abs(J - num).max() < 1e-7
    xn, _ = prior.rk4_step_with_jacobian(x)
    assert np.array_equal(xn, rk4_step(x)) or np.abs(xn - rk4_step(x)).max() < 1e-12
