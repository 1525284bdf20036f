"""ctypes binding of libfermiflow_hip.so (the C ABI declared in include/fermiflow.h).

The library is the product: if it is missing, or a tensor is not a contiguous fp64 CUDA tensor, the
calls raise -- there is no CPU or PyTorch fallback anywhere in this package.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FERMIFLOW_LIB") or os.path.join(_HERE, "libfermiflow_hip.so")   # env: A/B builds in tools/
_LIB = None

ABI_VERSION = 109      # ff_version() of the library this binding was written against (include/fermiflow.h)

SYMBOLS = [
    "ff_version", "ff_last_error", "ff_fermion_states", "ff_slater_logabsdet_fwd", "ff_slater_logabsdet_bwd", "ff_logprob",
    "ff_mcmc_sample_noise", "ff_mcmc_sample", "ff_mcmc_continue", "ff_rng_fill", "ff_mlp_eval", "ff_backflow_v_div", "ff_potential", "ff_radial_table_bytes", "ff_radial_table_build",
    "ff_cnf_generate", "ff_cnf_delta_logp", "ff_cnf_adjoint_workspace_bytes", "ff_cnf_adjoint", "ff_cnf_adjoint_energy", "ff_reduce_energy", "ff_energy_finish", "ff_stream_delay",
    "ff_eloc_workspace_bytes", "ff_eloc", "ff_eloc_sensitivities", "ff_eloc_finish", "ff_reduce_moments", "ff_state_sums", "ff_beta_buffer_doubles", "ff_beta_state_partials", "ff_beta_finish", "ff_logprob3d", "ff_mcmc_sample_noise3d", "ff_mcmc_sample3d", "ff_eloc_finish3d", "ff_backflow_v_div_f32", "ff_walker_order_workspace_bytes", "ff_walker_order", "ff_set_kernel_family", "ff_set_sens_precision", "ff_shutdown", "ff_walker_order_mean", "ff_energy_estimate_workspace_bytes", "ff_energy_estimate", "ff_mlp_eval_nd", "ff_backflow_vjp", "ff_eloc_nd", "ff_eloc_nd_workspace_bytes", "ff_rng_fill3d", "ff_walker_schedule", "ff_scale_counts", "ff_comm_unique_id", "ff_comm_init", "ff_comm_allreduce", "ff_comm_destroy", "ff_adam_step",
]


class FFNet(C.Structure):
    _fields_ = [("He", C.c_int32), ("ew1", C.c_void_p), ("eb1", C.c_void_p), ("ew2", C.c_void_p),
                ("Hm", C.c_int32), ("mw1", C.c_void_p), ("mb1", C.c_void_p), ("mw2", C.c_void_p),
                ("radial_table", C.c_void_p)]


class FFOde(C.Structure):
    _fields_ = [("t0", C.c_double), ("t1", C.c_double), ("rtol", C.c_double), ("atol", C.c_double),
                ("max_steps", C.c_int32), ("walker_cost", C.c_void_p), ("walker_order", C.c_void_p),
                ("walker_h_init", C.c_void_p), ("walker_h_scale", C.c_double), ("walker_h_out", C.c_void_p),
                ("walker_class", C.c_void_p), ("sens_tol", C.c_double), ("walker_h_scale_loose", C.c_double), ("sens_tol_class", C.c_int32),
                ("walker_h_uniform", C.c_int32), ("heavy_class", C.c_int32), ("heavy_tol", C.c_double), ("sum_weight", C.c_double),
                ("compact_finish", C.c_int32), ("after_main_event", C.c_void_p), ("walker_h_equal", C.c_int32)]


def lib():
    """Load the HIP library (once). Raises RuntimeError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP kernels first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C fermiflow_amd/csrc). fermiflow_amd has no CPU fallback.")
        _LIB = C.CDLL(LIB_PATH)
        if _LIB.ff_version() != ABI_VERSION:      # a stale build would read FFOde with the wrong layout
            v, _LIB = _LIB.ff_version(), None
            raise RuntimeError(f"{LIB_PATH} has ABI version {v}, this binding expects {ABI_VERSION}: rebuild "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        _LIB.ff_last_error.restype = C.c_char_p
        _LIB.ff_eloc_workspace_bytes.restype = C.c_size_t
        _LIB.ff_eloc_nd_workspace_bytes.restype = C.c_size_t
        _LIB.ff_cnf_adjoint_workspace_bytes.restype = C.c_size_t
        _LIB.ff_radial_table_bytes.restype = C.c_size_t
        _LIB.ff_walker_order_workspace_bytes.restype = C.c_size_t
        _LIB.ff_fermion_states.restype = C.c_int64
        _LIB.ff_beta_buffer_doubles.restype = C.c_size_t
        _LIB.ff_energy_estimate_workspace_bytes.restype = C.c_size_t
    return _LIB


def check(status, what):
    if status == 0:
        return
    msg = lib().ff_last_error().decode()
    if status == 1:
        raise ValueError(f"{what}: {msg}")
    if status == 2:
        raise NotImplementedError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: HIP failure: {msg}")


def dev(t, dtype=torch.float64, name="tensor"):
    """Validate a tensor for the C ABI and return it (contiguous)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a ROCm device (got {t.device}); fermiflow_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def i64(v):
    return C.c_int64(int(v))


def f64(v):
    return C.c_double(float(v))


RADIAL_MODE = os.environ.get("FERMIFLOW_RADIAL", "table")   # "table" (default) | "exact": how the ODE kernels evaluate eta, mu


class Net:
    """Device pointers of the backflow's two scalar MLPs (keeps the tensors alive).  With radial="table" the
    per-launch radial table (csrc/ff_radial.h) is built here, on the current stream, for exactly these weights."""

    def __init__(self, eta, mu=None, radial=None):
        self.t = []

        def three(m):
            ws = (dev(m.fc1.weight.detach().reshape(-1)), dev(m.fc1.bias.detach()), dev(m.fc2.weight.detach().reshape(-1)))
            self.t.extend(ws)
            return ws
        e = three(eta)
        m = three(mu) if mu is not None else (None, None, None)
        self.He = e[0].numel()
        self.Hm = m[0].numel() if mu is not None else 0
        self.c = FFNet(self.He, ptr(e[0]), ptr(e[1]), ptr(e[2]), self.Hm, ptr(m[0]), ptr(m[1]), ptr(m[2]), None)
        self.device = e[0].device
        if (radial or RADIAL_MODE) == "table":
            tab = torch.empty(lib().ff_radial_table_bytes() // 8, dtype=torch.float64, device=self.device)
            check(lib().ff_radial_table_build(stream(), C.byref(self.c), ptr(tab)), "ff_radial_table_build")
            self.t.append(tab)
            self.c.radial_table = tab.data_ptr()

    @property
    def nparams(self):
        return 3 * self.He + 3 * self.Hm

    def ref(self):
        return C.byref(self.c)


def ode(t0, t1, rtol, atol, max_steps=0, walker_cost=None, walker_order=None, walker_h_init=None, walker_h_scale=1.0,
        walker_h_out=None, walker_h_uniform=False, walker_class=None, sens_tol=1.0, sens_tol_class=0,
        walker_h_scale_loose=0.0, heavy_class=0, heavy_tol=0.0, sum_weight=0.0, compact_finish=False, after_main_event=None, walker_h_equal=False):
    """ff_ode; walker_cost (out) / walker_order (in): optional int32 tensors of length B (scheduling aids);
    walker_h_init (in) / walker_h_out (out): optional float64 tensors of length B (step-size warm start);
    walker_h_uniform: walker_h_init is a 1-element tensor, the first step of every walker;
    heavy_class / heavy_tol / sum_weight: routing threshold and tolerances of the local-energy pass (0: library defaults 12 (16 at 12 coordinates), 0.3, 4;
    heavy_class < 0: no routing); compact_finish: ff_eloc_nd finishes walkers in the one-walker-per-workgroup kernels' epilogue
    (compact workspace beyond 24 coordinates; include/fermiflow.h); walker_h_equal: flow and adjoint passes round the opening step
    walker_h_init x walker_h_scale down to equal steps of the interval."""
    for name, tns, dt in (("walker_cost", walker_cost, torch.int32), ("walker_order", walker_order, torch.int32),
                          ("walker_h_init", walker_h_init, torch.float64), ("walker_h_out", walker_h_out, torch.float64),
                          ("walker_class", walker_class, torch.int32)):
        if tns is not None and not (tns.dtype == dt and tns.is_contiguous() and tns.is_cuda):
            raise ValueError(f"{name} must be a contiguous {dt} device tensor")
    p = lambda t: t.data_ptr() if t is not None else None
    return FFOde(float(t0), float(t1), float(rtol), float(atol), int(max_steps), p(walker_cost), p(walker_order),
                 p(walker_h_init), float(walker_h_scale), p(walker_h_out), p(walker_class), float(sens_tol), float(walker_h_scale_loose),
                 int(sens_tol_class), int(bool(walker_h_uniform)), int(heavy_class), float(heavy_tol), float(sum_weight), int(bool(compact_finish)),
                 (int(after_main_event.cuda_event) or None) if after_main_event is not None else None, int(bool(walker_h_equal)))
