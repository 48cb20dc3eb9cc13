"""NMPC problem constants under the reference's attribute names (params/nmpc_params.py:5-43), so that code written
against `params.nmpc_params` (N_node, th_pred, Qp_xy, xr_list_index ...) keeps working.  Values are grouped by role;
the module attributes are generated from the groups."""
from . import fhnp_params as _vehicle

_HORIZON = {"N_node": 20, "T_horizon": 2, "ts_nmpc": 0.02}                    # shooting nodes, horizon [s], control period [s]
_DIMS = {"n_states": 10, "n_controls": 4}
_BOUNDS = {"w": (-6, 6), "c": (0, _vehicle.c_max), "v": (-20, 20)}             # body rates [rad/s], thrust accel., velocity [m/s]
_WEIGHTS = {"Qp_xy": 300, "Qp_z": 400, "Qv_xy": 10, "Qv_z": 10, "Qq_xy": 10, "Qq_z": 100, "Rw": 10, "Rc": 5}

globals().update(_HORIZON)
globals().update(_DIMS)
globals().update(_WEIGHTS)
for _name, (_lo, _hi) in _BOUNDS.items():
    globals()[_name + "_min"], globals()[_name + "_max"] = _lo, _hi
gravity, mass = _vehicle.gravity, _vehicle.mass
th_pred = _HORIZON["T_horizon"] / _HORIZON["N_node"]                          # shooting interval [s]

# The reference generator keeps one point per control period over the horizon and hands out every
# (th_pred / ts_nmpc)-th of them (nmpc_params.py:40-43): 101 points, stride 5, at the reference values.
_per_interval = th_pred / _HORIZON["ts_nmpc"]
if abs(_per_interval - round(_per_interval)) > 1e-6:
    raise ValueError("th_pred must be an integer multiple of ts_nmpc")
long_list_size = int(round(_per_interval)) * _HORIZON["N_node"] + 1
xr_list_index = slice(0, long_list_size, int(round(_per_interval)))
