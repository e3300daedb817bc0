"""ctypes binding of the C-ABI (include/mazestep.h) — libmazestep.so.

There is no CPU path: if the HIP library is missing, or no GPU is visible when an
environment is created, the call raises.  (`load(require_gpu=False)` is only for
checking that the library loads and exports every declared symbol.)
"""
import ctypes as C
import os

from mujoco_maze_amd.model import MzModel

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB_PATH = os.path.join(_HERE, "csrc", "libmazestep.so")
# A/B timing of experiment builds (tools/exp_*.sh) only: the override is announced on stderr when the library is loaded and
# recorded in bench.py's JSON line (config.library), so a stale variable cannot swap the stepper silently
# — and it is honoured only together with MZ_DEBUG=1 (a developer's shell), never on its own
LIB_PATH = (os.environ.get("MZ_LIBMAZESTEP_EXPERIMENT") if os.environ.get("MZ_DEBUG") == "1" else None) or DEFAULT_LIB_PATH

# every entry point declared in include/mazestep.h
SYMBOLS = [
    "mz_abi_version", "mz_model_sizeof", "mz_create", "mz_destroy", "mz_last_error", "mz_num_envs", "mz_obs_dim", "mz_nq",
    "mz_nv", "mz_nu", "mz_set_option", "mz_reset", "mz_set_state", "mz_get_state", "mz_step", "mz_get_status",
    "mz_debug_forward", "mz_last_kernel_ms", "mz_read_phase_cycles", "mz_bind_final_obs", "mz_debug_task_eval", "mz_debug_detect", "mz_read_wave_cycles", "mz_bind_record",
    "mz_set_goals", "mz_read_wave_phase_cycles", "mz_get_info", "mz_bind_env_goals",
]

_lib = None


class MazeStepError(RuntimeError):
    pass


def load():
    """Load libmazestep.so (built in-tree by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MazeStepError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C mujoco_maze_amd/csrc). mujoco_maze_amd has no CPU fallback.")
    # torch first: its wheel carries its own HIP runtime, and the device buffers this library works on are torch tensors — if
    # libmazestep.so were loaded before torch, the process would hold the system's libamdhip64 as well and the second runtime to
    # initialise sees no device (found on the GPU box: build() loading the library ahead of smoke()'s `import torch`)
    import torch  # noqa: F401

    if LIB_PATH != DEFAULT_LIB_PATH:
        import sys

        print(f"mujoco_maze_amd: MZ_LIBMAZESTEP_EXPERIMENT is set — loading {LIB_PATH} instead of {DEFAULT_LIB_PATH}", file=sys.stderr)
    lib = C.CDLL(LIB_PATH)
    vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
    lib.mz_abi_version.restype = C.c_int
    lib.mz_model_sizeof.restype = u64
    lib.mz_create.restype = vp
    lib.mz_create.argtypes = [C.POINTER(MzModel), i32, i32, C.c_char_p, i32]
    lib.mz_destroy.argtypes = [vp]
    lib.mz_last_error.restype = C.c_char_p
    lib.mz_last_error.argtypes = [vp]
    for name in ("mz_num_envs", "mz_obs_dim", "mz_nq", "mz_nv", "mz_nu"):
        getattr(lib, name).restype = i32
        getattr(lib, name).argtypes = [vp]
    lib.mz_set_option.restype = i32
    lib.mz_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.mz_bind_env_goals.restype = i32
    lib.mz_bind_env_goals.argtypes = [vp, vp, vp]
    lib.mz_get_info.restype = i32
    lib.mz_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    lib.mz_reset.restype = i32
    lib.mz_reset.argtypes = [vp, vp, u64, vp, vp]
    lib.mz_set_state.restype = i32
    lib.mz_set_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mz_get_state.restype = i32
    lib.mz_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mz_step.restype = i32
    lib.mz_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mz_get_status.restype = i32
    lib.mz_get_status.argtypes = [vp, vp, vp]
    lib.mz_debug_forward.restype = i32
    lib.mz_debug_forward.argtypes = [vp, vp, vp, vp, vp]
    lib.mz_bind_final_obs.restype = i32
    lib.mz_bind_final_obs.argtypes = [vp, vp]
    lib.mz_bind_record.restype = i32
    lib.mz_bind_record.argtypes = [vp, vp]
    lib.mz_set_goals.restype = i32
    lib.mz_set_goals.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.mz_debug_task_eval.restype = i32
    lib.mz_debug_task_eval.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.mz_debug_detect.restype = i32
    lib.mz_debug_detect.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    lib.mz_read_phase_cycles.restype = i32
    lib.mz_read_phase_cycles.argtypes = [vp, vp]
    lib.mz_read_wave_cycles.restype = i32
    lib.mz_read_wave_cycles.argtypes = [vp, vp, i32]
    lib.mz_read_wave_phase_cycles.restype = i32
    lib.mz_read_wave_phase_cycles.argtypes = [vp, vp, i32]
    lib.mz_last_kernel_ms.restype = C.c_double
    lib.mz_last_kernel_ms.argtypes = [vp]
    if lib.mz_model_sizeof() != C.sizeof(MzModel):
        raise MazeStepError("mz_model layout mismatch between include/mazestep.h and mujoco_maze_amd/model.py")
    _lib = lib
    return lib


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.mz_last_error(handle)
        raise MazeStepError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
