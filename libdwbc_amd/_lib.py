"""ctypes binding of libdwbc_hip.so -- the C-ABI declared in include/dwbc_batch.h.

There is deliberately no fallback: if the HIP library is missing or there is no GPU, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DWBC_TIMED=1 selects the diagnostic build with in-kernel stage stamps (never used for reported numbers)
_VARIANT = os.environ.get("DWBC_LIB_VARIANT")  # development A/B builds (csrc/Makefile target `experiment`)
LIB_PATH = os.path.join(_HERE, "libdwbc_hip_timed.so" if os.environ.get("DWBC_TIMED") == "1" else (f"libdwbc_hip_{_VARIANT}.so" if _VARIANT else "libdwbc_hip.so"))

# every symbol include/dwbc_batch.h declares: (name, restype, argtypes)
_vp, _i, _d, _cp = C.c_void_p, C.c_int, C.c_double, C.c_char_p
SYMBOLS = [
    ("dwbc_last_error", _cp, []),
    ("dwbc_device_count", _i, []),
    ("dwbc_model_create_from_urdf", _vp, [_cp, _i]),
    ("dwbc_model_create_from_arrays", _vp, [_i] + [_vp] * 7),
    ("dwbc_model_destroy", None, [_vp]),
    ("dwbc_model_num_links", _i, [_vp]),
    ("dwbc_model_system_dof", _i, [_vp]),
    ("dwbc_model_total_mass", _d, [_vp]),
    ("dwbc_model_link_id", _i, [_vp, _cp]),
    ("dwbc_model_link_name", _cp, [_vp, _i]),
    ("dwbc_model_get_arrays", _i, [_vp] + [_vp] * 7),
    ("dwbc_model_delete_link", _vp, [_vp, _i]),
    ("dwbc_model_add_link", _vp, [_vp, _i, C.c_char_p, _i, _vp, _vp, _vp, C.c_double, _vp, _vp]),
    ("dwbc_model_change_link_to_fixed_joint", _vp, [_vp, _i]),
    ("dwbc_model_change_link_inertia", _vp, [_vp, _i, _vp, _vp, C.c_double]),
    ("dwbc_batch_create", _vp, [_vp, _i, _i, _i]),
    ("dwbc_batch_destroy", None, [_vp]),
    ("dwbc_batch_size", _i, [_vp]),
    ("dwbc_batch_add_contact", _i, [_vp, _i, _i, _vp, _d, _d, _d, _d]),
    ("dwbc_batch_clear_contacts", _i, [_vp]),
    ("dwbc_batch_add_task", _i, [_vp, _i, _i, _i, _vp]),
    ("dwbc_batch_add_custom_task", _i, [_vp, _i, _i]),
    ("dwbc_batch_set_custom_task", _i, [_vp, _i, _vp, _vp]),
    ("dwbc_batch_clear_tasks", _i, [_vp]),
    ("dwbc_batch_set_task_gain", _i, [_vp, _i, _i] + [_vp] * 6),
    ("dwbc_batch_set_trajectory", _i, [_vp, _i, _i, _vp]),
    ("dwbc_batch_set_control_time", _i, [_vp, _vp]),
    ("dwbc_batch_set_torque_limit", _i, [_vp, _vp]),
    ("dwbc_batch_fstar_size", _i, [_vp]),
    ("dwbc_batch_task_dof", _i, [_vp, _i]),
    ("dwbc_batch_set_state", _i, [_vp, _vp, _vp, _vp]),
    ("dwbc_batch_set_contact", _i, [_vp, _vp]),
    ("dwbc_batch_set_max_active_contacts", _i, [_vp, _i]),
    ("dwbc_batch_max_active_contacts", _i, [_vp]),
    ("dwbc_batch_set_fstar", _i, [_vp, _i, _vp]),
    ("dwbc_batch_copy_kinematics", _i, [_vp, _vp]),
    ("dwbc_batch_bind_device", _i, [_vp, _i, _vp]),
    ("dwbc_batch_set_stream", _i, [_vp, _vp]),
    ("dwbc_batch_enable_dump", _i, [_vp, _i]),
    ("dwbc_batch_solve", _i, [_vp, C.c_uint]),
    ("dwbc_batch_sync", _i, [_vp]),
    ("dwbc_batch_time_solves", _i, [_vp, C.c_uint, _i, C.POINTER(C.c_float)]),
    ("dwbc_batch_get", _i, [_vp, _i, _vp, C.c_size_t]),
    ("dwbc_batch_field_bytes", C.c_size_t, [_vp, _i]),
    ("dwbc_batch_launch_info", _i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    ("dwbc_batch_kernel_name", C.c_char_p, [_vp]),
    # generic hierarchical-QP class + LQP configurator
    ("dwbc_hqp_create", _vp, [_i, _i, _i, _i, _i]),
    ("dwbc_hqp_destroy", None, [_vp]),
    ("dwbc_hqp_add_hierarchy", _i, [_vp, _i, _i]),
    ("dwbc_hqp_clear", _i, [_vp]),
    ("dwbc_hqp_update_constraint_matrix", _i, [_vp, _i, _vp, _vp, _vp, _vp]),
    ("dwbc_hqp_update_cost_matrix", _i, [_vp, _i, _vp, _vp]),
    ("dwbc_hqp_normalize_constraint_matrix", _i, [_vp, _i]),
    ("dwbc_hqp_update_constraint_weight", _i, [_vp, _i, _vp, _vp]),
    ("dwbc_hqp_set_answer", _i, [_vp, _i, _vp, _vp]),
    ("dwbc_hqp_prepare", _i, [_vp]),
    ("dwbc_hqp_solve_first", _i, [_vp, _i]),
    ("dwbc_hqp_solve_sequential", _i, [_vp, _i]),
    ("dwbc_hqp_num_levels", _i, [_vp]),
    ("dwbc_hqp_field_bytes", C.c_size_t, [_vp, _i, _i]),
    ("dwbc_hqp_get", _i, [_vp, _i, _i, _vp, C.c_size_t]),
    ("dwbc_batch_configure_lqp", _i, [_vp, _vp]),
    ("dwbc_batch_lqp_torque", _i, [_vp, _vp, _vp]),
    ("dwbc_batch_solve_jacc", _i, [_vp, _vp, _i]),
    ("dwbc_batch_get_jacc", _i, [_vp, _i, _i, _vp, C.c_size_t]),
    ("dwbc_batch_reduced_dims", _i, [_vp, _vp, _vp]),
    ("dwbc_batch_configure_lqp_r", _i, [_vp, _vp]),
    ("dwbc_batch_configure_lqp_r_nc", _i, [_vp, _vp, _vp, _i]),
    ("dwbc_batch_solve_jacc_r", _i, [_vp, _vp, _i]),
    ("dwbc_batch_solve_jacc_r_nc", _i, [_vp, _vp, _i, _i]),
    ("dwbc_batch_get_jacc_nc", _i, [_vp, _i, _vp, C.c_size_t]),
    ("dwbc_batch_host_ptr", _vp, [_vp, _i]),
]

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  libdwbc_amd has no CPU fallback."
            )
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  When
        # torch is going to share device buffers / streams with this library it must be the first to load the
        # runtime, otherwise torch.cuda fails with "No HIP GPUs are available".  torch is optional.
        if os.environ.get("DWBC_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return load().dwbc_last_error().decode()
