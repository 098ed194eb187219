"""ctypes binding of liblrnde.so (include/lrnde.h).  No fallback: if the HIP
library is missing or fails to load, importing this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblrnde.so")

STATUS = {0: "Success", 1: "MaxIters", 2: "DtLessThanMin", 3: "DtNaN", 4: "BadArg", 5: "Capacity",
          6: "HipError", 7: "NcclError", 8: "Unsupported"}
ACT = {"identity": 0, "tanh": 1, "gelu": 2}
REG_TYPE = {"error_estimate": 0, "stiffness_estimate": 1}
MODE = {"none": 0, "unbiased": 1, "biased": 2}
DTYPE = {"f32": 0, "bf16": 1, "f32_split": 2}


class ModelDesc(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("hidden_dim", C.c_int32), ("time_dep", C.c_int32),
                ("act", C.c_int32)]


SRI_FIELDS = ("a021 a031 a032 a041 a042 a043 a121 a131 a132 a141 a142 a143 "
              "b021 b031 b032 b041 b042 b043 b121 b131 b132 b141 b142 b143 "
              "c02 c03 c04 c11 c12 c13 c14 alpha1 alpha2 alpha3 alpha4 "
              "beta11 beta12 beta13 beta14 beta21 beta22 beta23 beta24 beta31 beta32 beta33 beta34 beta41 beta42 beta43 beta44").split()


class SriTableau(C.Structure):
    """lrnde_sri_tableau: the fields of StochasticDiffEq's FourStageSRIConstantCache in the order src/perform_step.jl:51-55 unpacks them"""
    _fields_ = [(n, C.c_float) for n in SRI_FIELDS]


class ConvDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("hidden", C.c_int32),
                ("act", C.c_int32), ("bn_train", C.c_int32), ("compute_dtype", C.c_int32), ("bn_eps", C.c_float)]


class SdeAdaptOpts(C.Structure):
    _fields_ = [("abstol", C.c_float), ("reltol", C.c_float), ("delta", C.c_float), ("dt0", C.c_float), ("gamma", C.c_float),
                ("qmin", C.c_float), ("qmax", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("maxiters", C.c_int32)]


class SolveOpts(C.Structure):
    _fields_ = [("abstol", C.c_float), ("reltol", C.c_float), ("maxiters", C.c_int32),
                ("save_start", C.c_int32), ("save_everystep", C.c_int32), ("exact_pow", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("retcode", C.c_int32), ("nf", C.c_int32), ("naccept", C.c_int32),
                ("nreject", C.c_int32), ("iters", C.c_int32), ("nsaved", C.c_int32),
                ("t_final", C.c_float), ("dt_final", C.c_float), ("eest_last", C.c_float),
                ("dt_init", C.c_float)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TraceRow(C.Structure):
    _fields_ = [("t", C.c_float), ("dt", C.c_float), ("eest", C.c_float), ("accepted", C.c_int32)]


# every symbol include/lrnde.h and include/lrnde_hooks.h declare: (name, restype, argtypes)
_vp, _fp, _i32, _f = C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_float
SYMBOLS = [
    ("lrnde_create", C.c_int, [C.POINTER(_vp), C.POINTER(ModelDesc), C.c_int, _vp]),
    ("lrnde_destroy", C.c_int, [_vp]),
    ("lrnde_last_error", C.c_char_p, [_vp]),
    ("lrnde_param_count", C.c_size_t, [C.POINTER(ModelDesc)]),
    ("lrnde_version", C.c_char_p, []),
    ("lrnde_set_params", C.c_int, [_vp, _vp, C.c_size_t]),
    ("lrnde_set_solver", C.c_int, [_vp, _i32]),
    ("lrnde_rhs", C.c_int, [_vp, _vp, _f, _i32, _vp]),
    ("lrnde_init_dt", C.c_int, [_vp, _vp, _i32, _f, _f, _f, _f, _vp, _fp]),
    ("lrnde_perform_step", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _vp, _vp, _fp, _fp, _fp]),
    ("lrnde_solve", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _fp, _i32, _vp, _fp, _i32,
                              C.POINTER(Stats), C.POINTER(TraceRow), _i32]),
    ("lrnde_node_forward", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _vp,
                                     _fp, C.POINTER(_i32), C.POINTER(Stats), _fp]),
    ("lrnde_comm_unique_id", C.c_int, [_vp]),
    ("lrnde_comm_init", C.c_int, [_vp, _vp, _i32, _i32]),
    ("lrnde_comm_destroy", C.c_int, [_vp]),
    ("lrnde_comm_count", C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    ("lrnde_bench_exchange", C.c_int, [_vp, _i32, _i32, _fp]),
    ("lrnde_local_comm_create", C.c_int, [C.POINTER(_vp), _i32]),
    ("lrnde_local_comm_destroy", C.c_int, [_vp]),
    ("lrnde_comm_init_local", C.c_int, [_vp, _vp, _i32]),
    ("lrnde_sde_create", C.c_int, [C.POINTER(_vp), C.POINTER(ModelDesc), _i32, C.c_int, _vp]),
    ("lrnde_sde_destroy", C.c_int, [_vp]),
    ("lrnde_sde_last_error", C.c_char_p, [_vp]),
    ("lrnde_sde_set_params", C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t]),
    ("lrnde_sde_euler_heun_step", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _f, _vp, _fp, _fp]),
    ("lrnde_sde_rkmil_step", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _vp, _fp, _fp]),
    ("lrnde_sde_solve_fixed", C.c_int, [_vp, _i32, _vp, _vp, _i32, _f, _f, _i32, _f, _f, _f, _vp, _fp, _fp]),
    ("lrnde_sde_solve_adaptive", C.c_int, [_vp, _vp, _vp, _i32, _i32, _f, _f, C.POINTER(SdeAdaptOpts), _vp, C.POINTER(Stats),
                                           C.POINTER(TraceRow), _i32]),
    ("lrnde_sde_solve_fixed_backward", C.c_int, [_vp, _vp, _vp, _vp, _i32, _f, _f, _i32, _vp, _vp, _vp, _vp]),
    ("lrnde_sde_euler_heun_reg_grad", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _f, _vp, _vp, _fp]),
    ("lrnde_sde_node_forward_record", C.c_int, [_vp, _vp, _vp, _i32, _i32, _f, _f, C.POINTER(SdeAdaptOpts), _i32, _f, _vp, _i32, _fp, _i32,
                                                _vp, _fp, _i32, C.POINTER(_i32), _fp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(Stats), _fp]),
    ("lrnde_sde_node_backward_recorded", C.c_int, [_vp, _i32, _vp, _i32, _f, _vp, _vp, _vp]),
    ("lrnde_sde_sri_step_backward", C.c_int, [_vp, C.POINTER(SriTableau), _vp, _vp, _vp, _i32, _f, _f, _f, _f, _f, _vp, _f, _vp, _vp, _vp, _fp]),
    ("lrnde_record_generation", C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    ("lrnde_conv_record_generation", C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    ("lrnde_sde_record_generation", C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    ("lrnde_sde_solve_fixed_backward_rkmil", C.c_int, [_vp, _vp, _vp, _vp, _i32, _f, _f, _i32, _vp, _vp, _vp, _vp]),
    ("lrnde_sde_rkmil_reg_grad", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _vp, _vp, _fp]),
    ("lrnde_sde_sri_step", C.c_int, [_vp, C.POINTER(SriTableau), _vp, _vp, _vp, _i32, _f, _f, _f, _f, _f, _vp, _fp, _fp]),
    ("lrnde_vjp", C.c_int, [_vp, _vp, _f, _vp, _i32, _vp, _vp]),
    ("lrnde_step_reg_grad", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _i32, _vp, _fp]),
    ("lrnde_node_backward", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _vp, _f, _vp, _vp,
                                      C.POINTER(Stats), C.POINTER(Stats)]),
    ("lrnde_node_forward_record", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _vp,
                                            _fp, C.POINTER(_i32), C.POINTER(Stats), _fp]),
    ("lrnde_node_backward_recorded", C.c_int, [_vp, _i32, _vp, _f, _vp, _vp, C.POINTER(Stats)]),
    ("lrnde_node_forward_record_ts", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _fp, _i32, _vp, _fp, _i32,
                                               C.POINTER(_i32), _fp, C.POINTER(_i32), C.POINTER(Stats), _fp]),
    ("lrnde_node_backward_recorded_ts", C.c_int, [_vp, _i32, _vp, _i32, _f, _vp, _vp, C.POINTER(Stats)]),
    ("lrnde_opt_update", C.c_int, [_i32, _vp, _vp, _vp, _vp, C.c_size_t, _f, _f, _f, _f, _i32, _f, C.c_int, _vp]),
    ("lrnde_classifier_ce", C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _fp, _vp, _vp, _vp]),
    ("lrnde_node_forward_record_ce", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _vp,
                                               _fp, C.POINTER(_i32), C.POINTER(Stats), _fp, _vp, _i32, _vp, _fp, _vp, _vp, _vp]),
    ("lrnde_conv_param_count", C.c_size_t, [C.POINTER(ConvDesc)]),
    ("lrnde_conv_create", C.c_int, [C.POINTER(_vp), C.POINTER(ConvDesc), C.c_int, _vp]),
    ("lrnde_conv_destroy", C.c_int, [_vp]),
    ("lrnde_conv_last_error", C.c_char_p, [_vp]),
    ("lrnde_conv_set_params", C.c_int, [_vp, _vp, C.c_size_t]),
    ("lrnde_conv_set_bn_state", C.c_int, [_vp, _vp, C.c_size_t]),
    ("lrnde_conv_get_bn_state", C.c_int, [_vp, _vp, C.c_size_t]),
    ("lrnde_conv_set_bn_mode", C.c_int, [_vp, _i32]),
    ("lrnde_conv_rhs", C.c_int, [_vp, _vp, _f, _i32, _vp]),
    ("lrnde_conv_init_dt", C.c_int, [_vp, _vp, _i32, _f, _f, _f, _f, _vp, _fp]),
    ("lrnde_conv_perform_step", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _vp, _vp, _fp, _fp, _fp]),
    ("lrnde_conv_solve", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _fp, _i32, _vp, _fp, _i32,
                                   C.POINTER(Stats), C.POINTER(TraceRow), _i32]),
    ("lrnde_conv_node_forward", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _vp,
                                          _fp, C.POINTER(_i32), C.POINTER(Stats), _fp]),
    ("lrnde_conv_vjp", C.c_int, [_vp, _vp, _f, _vp, _i32, _vp, _vp]),
    ("lrnde_conv_step_reg_grad", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _i32, _vp, _fp]),
    ("lrnde_conv_node_backward", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _vp, _f, _vp, _vp,
                                           C.POINTER(Stats), C.POINTER(Stats)]),
    ("lrnde_conv_node_forward_record", C.c_int, [_vp, _vp, _i32, _f, _f, C.POINTER(SolveOpts), _i32, _i32, _f, _vp,
                                                 _fp, C.POINTER(_i32), C.POINTER(Stats), _fp]),
    ("lrnde_conv_node_backward_recorded", C.c_int, [_vp, _i32, _vp, _f, _vp, _vp, C.POINTER(Stats)]),
    ("lrnde_cifar_stem_param_count", C.c_size_t, []),
    ("lrnde_cifar_head_param_count", C.c_size_t, [_i32, _i32, _i32]),
    ("lrnde_cifar_stem_forward", C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    ("lrnde_cifar_stem_backward", C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    ("lrnde_cifar_head_ce", C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _fp, _vp, _vp, _vp]),
    ("lrnde_conv_bench_rhs", C.c_int, [_vp, _vp, _f, _i32, _i32, _fp]),
    ("lrnde_bench_step", C.c_int, [_vp, _vp, _vp, _i32, _f, _f, _f, _f, _i32, _fp]),
    ("lrnde_set_overlap", C.c_int, [_vp, _i32]),
    ("lrnde_set_reports", C.c_int, [_vp, _i32]),
    ("lrnde_set_option", C.c_int, [C.c_char_p, _i32]),
    ("lrnde_set_adjoint_trace", C.c_int, [_vp, C.POINTER(TraceRow), _i32]),
    ("lrnde_adjoint_trace_rows", C.c_int, [_vp, C.POINTER(_i32)]),
    ("lrnde_host_phases", C.c_int, [_vp, C.POINTER(C.c_double), _i32]),
    ("lrnde_last_solve_kernel_ms", C.c_int, [_vp, _fp, C.POINTER(_i32)]),
]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP library is the product path and there is no fallback. "
        "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C localregneuralde.jl_amd/csrc`).")

lib = C.CDLL(LIB_PATH)
for _name, _res, _args in SYMBOLS:
    _fn = getattr(lib, _name)  # AttributeError here = ABI drift, fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


class LrndeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lrnde status {code} ({STATUS.get(code, '?')}): {msg}")
        self.code = code


def check(ctx, rc):
    if rc != 0:
        msg = lib.lrnde_last_error(ctx)
        raise LrndeError(rc, msg.decode() if msg else "")


def set_option(name, value):
    """diagnostic hook lrnde_set_option (include/lrnde_hooks.h): override a process-wide switch by its environment-variable name"""
    rc = lib.lrnde_set_option(name.encode(), int(value))
    if rc != 0:
        raise LrndeError(rc, f"unknown option {name}")
