"""Can the downwash of the NEXT tick run beside the control-step kernel?  Two free-running streams, no dependencies between them:
   A: n control steps WITHOUT the fused downwash (force from a buffer) -- rti_kernel<3,4,false,20>, all of a CU's LDS, 320 registers
   B: n launches of the LDS-free downwash kernel (mlp_stream_kernel, <= 192 registers: can be co-resident)
each captured into its own hipGraph; timed alone, together, and against the fused single launch per tick."""
import ctypes as C
import sys
import time

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import torch

import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

B, T, n = 1024, 8, 200
dev = torch.device("cuda", 0)
ticks = []
for t in range(T):
    b = synth.make_batch(B, seed=1, downwash=True, t0=0.02 * t)
    ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
eng = ndp.BatchedNMPC(B, disturbance=True)
eng2 = ndp.BatchedNMPC(B, disturbance=True)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
f = torch.zeros(B, 21, 3, dtype=torch.float32, device=dev)
f2 = torch.zeros(B, 21, 3, dtype=torch.float32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def rti(i, s, e=eng):
    d = ticks[i % T]
    e.update_device(d["x0"], d["xr"], d["ur"], u0, f=f, stream=s)


def fused(i, s):
    d = ticks[i % T]
    eng2.update_device(d["x0"], d["xr"], d["ur"], u0, other=d["other"], ego_xy=d["ego_xy"], stream=s)


def mlp(i, s):
    d = ticks[i % T]
    assert eng._lib.ndp_debug_downwash_stream_device(eng._h, p(d["other"]), p(d["xr"]), p(d["ego_xy"]), p(f2), C.c_void_p(s.cuda_stream)) == 0


def mlp_lds(i, s):
    d = ticks[i % T]
    eng.downwash_device(d["other"], d["xr"], f2, ego_xy=d["ego_xy"], stream=s)


def graph_of(fn, s):
    eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=s)
    eng2.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=s)
    for i in range(8):
        fn(i, s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
        for i in range(n):
            fn(i, s)
    with torch.cuda.stream(s):
        g.replay()
    torch.cuda.synchronize()
    return g, s


def timed(graphs, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g, st in graphs:
            with torch.cuda.stream(st):       # a graph replays on the CURRENT stream: each on its own
                g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / n * 1e6


# parity of the LDS-free kernel with the LDS one
mlp_lds(0, sa)
torch.cuda.synchronize()
ref = f2.clone()
mlp(0, sa)
torch.cuda.synchronize()
print("LDS-free downwash kernel vs LDS form: max |df| %.3e (|f| up to %.2f)" % (float((f2 - ref).abs().max()), float(ref.abs().max())))
gA, gB, gF, gL = graph_of(rti, sa), graph_of(mlp, sb), graph_of(fused, sa), graph_of(mlp_lds, sb)
print("per tick (us): control step without downwash %.2f | LDS-free downwash kernel alone %.2f | LDS downwash kernel alone %.2f | fused single launch %.2f"
      % (timed([gA]), timed([gB]), timed([gL]), timed([gF])))
print("per tick (us): control step (stream A) and LDS-free downwash (stream B) TOGETHER %.2f   [perfect overlap = max of the two, none = their sum]"
      % timed([gA, gB]))
print("per tick (us): control step (stream A) and LDS downwash (stream B) together %.2f" % timed([gA, gL]))

# control: two instances of the SAME light kernel (160 registers, no LDS) on two streams -- do streams overlap here at all?
sc = torch.cuda.Stream(device=dev)
f3 = torch.zeros(B, 21, 3, dtype=torch.float32, device=dev)


def mlp_c(i, s):
    d = ticks[i % T]
    assert eng._lib.ndp_debug_downwash_stream_device(eng._h, p(d["other"]), p(d["xr"]), p(d["ego_xy"]), p(f3), C.c_void_p(s.cuda_stream)) == 0


gC = graph_of(mlp_c, sc)
print("per tick (us): LDS-free downwash on stream B and again on stream C together %.2f (alone %.2f each)" % (timed([gB, gC]), timed([gB])))
import os
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " streams:", hex(sa.cuda_stream), hex(sb.cuda_stream), hex(sc.cuda_stream))
