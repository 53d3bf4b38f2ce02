"""Public names of the reference's NLEVP module (src/NLEVP_exports.jl:1-17), device-backed."""
from .algebra import (exp_az, exp_delay, exp_pm, generate_1_gz, generate_exp_az, generate_gz_hz,  # noqa: F401
                      generate_Sigma_y_exp_ikx, generate_z_g_z, pow0, pow1, pow2, pow_, pow_a, tau_delay)
from .beyn import (beyn, compute_moment_matrices, gauss_points, generate_subspace, initialize_V, inpoly, moments2eigs,
                   pos_test, project, wn)  # noqa: F401
from .linopfam import (DeviceFamily, LinearOperatorFamily, Operator, Solution, Term, conv_radius, estimate_pol, pade,  # noqa: F401
                       pade_, poly_roots, polyval)
from .local_solvers import (count_poles_and_zeros, decode_error_flag, eigs, eigs_many, householder, householder_many,
                            householder_update, inveriter,  # noqa: F401
                            itsol_arpack_9999, itsol_arpack_exception, itsol_converged, itsol_impossible, itsol_isnan,
                            itsol_maxiter, itsol_singular_exception, itsol_slow_convergence, itsol_unknown, lancaster,
                            mslp, padesolve, rf2s, traceiter)
from .perturbation import (multi_indices_at_order, multinomcoeff, part2mult, partitions, perturb, perturb_,  # noqa: F401
                           perturb_disk, perturb_fast_, perturb_norm, perturb_norm_, weigh)
from .save import load_family, read_sol, save  # noqa: F401
from .solver import solve  # noqa: F401
