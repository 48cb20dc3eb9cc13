"""NMPCBodyRateController on MI355X -- same constructor / reset / update as the reference class
(ndp_nmpc/scripts/nmpc_ctl/nmpc_body_rate_ctl.py:20-112)."""

from ..batched import BatchedNMPC
from ..params import nmpc_params as CP
from ..solver_facade import SolverFacade


class NMPCBodyRateController(object):
    def __init__(self, is_build_acados=True, device=0):
        # is_build_acados is accepted for signature compatibility; there is nothing to generate or compile.
        # Deviations from the reference constructor: no os.chdir side effect, no ACADOS_SOURCE_DIR.
        self._engine = BatchedNMPC(batch=1, N=CP.N_node, disturbance=False, device=device)
        self.solver = SolverFacade(self._engine, disturbance=False)

    def reset(self, xr, ur):
        # reset x and u of the controller, which prevents warm-starting from the previous solution
        for i in range(self.solver.N):
            self.solver.set(i, "x", xr[i, :])
            self.solver.set(i, "u", ur[i, :])
        self.solver.set(self.solver.N, "x", xr[self.solver.N, :])

    def update(self, x0, xr, ur):
        # yref_k = [xr_k, ur_k] (terminal: state only), p_k = xr_k[6:10] (reference quaternion for the nonlinear quaternion
        # error): the reference's 2 (N + 1) solver.set calls (nmpc_body_rate_ctl.py:95-104) as array assignments
        self.solver.set_reference(xr, ur)

        u0 = self.solver.solve_for_x0(x0)  # feedback, take the first action

        if self.solver.status != 0:
            raise Exception("acados acados_ocp_solver returned status {}. Exiting.".format(self.solver.status))

        return u0
