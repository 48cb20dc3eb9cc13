"""NMPC problem constants (reference params/nmpc_params.py:5-43)."""
from . import fhnp_params as QD

gravity = QD.gravity
mass = QD.mass

N_node = 20
T_horizon = 2
ts_nmpc = 0.02  # control period, 50 Hz
th_pred = T_horizon / N_node  # shooting interval, s

n_states = 10
n_controls = 4

w_max = 6
w_min = -6
c_max = QD.c_max
c_min = 0
v_max = 20
v_min = -20

Qp_xy = 300
Qp_z = 400
Qv_xy = 10
Qv_z = 10
Qq_xy = 10
Qq_z = 100
Rw = 10
Rc = 5

# sliding-window indexing of the reference generator (nmpc_params.py:40-43)
long_list_size = int(th_pred * N_node / ts_nmpc) + 1
if th_pred * N_node / ts_nmpc - int(th_pred * N_node / ts_nmpc) > 1e-6:
    raise ValueError("th_pred must be an integer multiple of ts_nmpc")
xr_list_index = slice(0, long_list_size, int(th_pred / ts_nmpc))
