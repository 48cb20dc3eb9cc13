"""NDPNMPCBodyRateController on MI355X -- same constructor / reset / update(x0, xr, ur, f) as the
reference class (ndp_nmpc/scripts/ndp_nmpc_ctl/ndp_nmpc_body_rate_ctl.py:20-112)."""

from ..batched import BatchedNMPC
from ..params import nmpc_params as CP
from ..solver_facade import SolverFacade


class NDPNMPCBodyRateController(object):
    def __init__(self, is_build_acados=True, device=0):
        self._engine = BatchedNMPC(batch=1, N=CP.N_node, disturbance=True, device=device, load_mlp=False)
        self.solver = SolverFacade(self._engine, disturbance=True)

    def reset(self, xr, ur):
        for i in range(self.solver.N):
            self.solver.set(i, "x", xr[i, :])
            self.solver.set(i, "u", ur[i, :])
        self.solver.set(self.solver.N, "x", xr[self.solver.N, :])

    def update(self, x0, xr, ur, f):
        # yref_k = [xr_k, ur_k], p_k = [reference quaternion, disturbance force] (ndp_nmpc_body_rate_ctl.py:93-104): the
        # reference's 2 (N + 1) solver.set calls as array assignments
        self.solver.set_reference(xr, ur, f)

        u0 = self.solver.solve_for_x0(x0)  # feedback, take the first action

        if self.solver.status != 0:
            raise Exception("acados acados_ocp_solver returned status {}. Exiting.".format(self.solver.status))

        return u0
