"""ctypes binding of the C ABI in ``include/sgw.h`` (``sorrel_amd/csrc/libsgw.so``).

There is no CPU fallback: if the HIP library is missing, or a second HIP runtime
gets mapped next to PyTorch's, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_TYPES, MAX_CHANNELS, MAX_CHOICES, MAX_ACTIONS, MAX_AGENTS, MAX_LAYERS, MAX_DIM = 32, 16, 8, 16, 128, 7, 256
RULE_NONE, RULE_SPAWN, RULE_BECOME_IF = 0, 1, 2
NO_BORDER = 255
STEP_SWEEP, STEP_RANDOM_ACTIONS, STEP_NO_OBS, STEP_OBS_NEXT, STEP_OBS_NEXT_PACKED, STEP_NO_MOVE, STEP_OBS_AGENT_MAJOR = 1, 2, 4, 8, 16, 32, 64
OBS_POST_NONE, OBS_POST_CLIP255_DIV255 = 0, 1
AGENT_RULE_MOVE, AGENT_RULE_TAG, AGENT_RULE_CLEANUP = 0, 1, 2
ACTION_MOVE, ACTION_CLEAN, ACTION_ZAP = 0, 1, 2
OBS_F32, OBS_U8 = 0, 1
STATUS_OOB_MOVE, STATUS_BAD_ACTION, STATUS_BAD_TYPE, STATUS_BAD_POS = 1, 2, 4, 8
CAP_OBSERVE_ROWS, CAP_ACT, CAP_RESOLVE, CAP_OBS_AGENT_MAJOR, CAP_SWEEP_ROWS = 1, 2, 4, 8, 16
ACT_U8, ACT_I32, ACT_I64, ACT_QF32 = 0, 1, 2, 3
TAIL_NONE, TAIL_AGENT_IS_IT, TAIL_POSITION_TABLE = 0, 1, 2
OK, EINVAL, EHIP, ENOMEM = 0, -1, -2, -3


class SgwConfig(C.Structure):
    """Mirror of ``struct sgw_config`` (include/sgw.h); layout checked by tests/test_abi.py."""

    _fields_ = [
        ("height", C.c_int32), ("width", C.c_int32), ("layers", C.c_int32),
        ("num_agents", C.c_int32), ("vision_radius", C.c_int32),
        ("num_types", C.c_int32), ("num_channels", C.c_int32), ("num_actions", C.c_int32),
        ("agent_layer", C.c_int32), ("default_type", C.c_int32), ("fill_type", C.c_int32),
        ("obs_post", C.c_int32),
        ("action_dy", C.c_int8 * MAX_ACTIONS), ("action_dx", C.c_int8 * MAX_ACTIONS),
        ("agent_type", C.c_uint8 * MAX_AGENTS),
        ("type_value", C.c_double * MAX_TYPES),
        ("type_passable", C.c_uint8 * MAX_TYPES),
        ("type_rule", C.c_uint8 * MAX_TYPES),
        ("spawn_prob", C.c_double * MAX_TYPES),
        ("spawn_count", C.c_uint8 * MAX_TYPES),
        ("spawn_choice", (C.c_uint8 * MAX_CHOICES) * MAX_TYPES),
        ("appearance", (C.c_double * MAX_CHANNELS) * MAX_TYPES),
        ("layer_fill_type", C.c_uint8 * 8),
        ("layer_border_type", C.c_uint8 * 8),
        ("dense_prob", C.c_double),
        ("dense_count", C.c_uint8),
        ("dense_choice", C.c_uint8 * MAX_CHOICES),
        ("agent_rule", C.c_uint8), ("tag_it_type", C.c_uint8), ("tag_notit_type", C.c_uint8),
        ("reserved1", C.c_uint8 * 4),
        ("seed", C.c_uint64),
        ("first_env_id", C.c_uint64),
        ("num_envs", C.c_int64),
        ("tag_reward", C.c_double),
        ("rule_layer", C.c_int8 * MAX_TYPES), ("rule_become", C.c_uint8 * MAX_TYPES), ("rule_mask", C.c_uint32 * MAX_TYPES),
        ("action_kind", C.c_uint8 * MAX_ACTIONS), ("beam_radius", C.c_int32),
        ("clean_beam_type", C.c_uint8), ("zap_beam_type", C.c_uint8), ("reserved2", C.c_uint8 * 2),
        ("beam_block_mask", C.c_uint32), ("reward_total_factor", C.c_int32),
        ("grid_env_stride", C.c_int64),
    ]


FAMILY_WAVE, FAMILY_WORKGROUP, FAMILY_GENERIC = 1, 2, 3


class SgwPlan(C.Structure):
    """Mirror of ``struct sgw_plan_info`` (include/sgw.h): what ``sgw_create`` decides for a config, as pure host arithmetic."""

    _fields_ = [
        ("family", C.c_int32), ("lanes_per_env", C.c_int32), ("threads", C.c_int32), ("grid_blocks", C.c_int32),
        ("lds_bytes", C.c_int64),
        ("env_lds", C.c_int32), ("obs_stage", C.c_int32), ("stage_agents", C.c_int32), ("whole_env_burst", C.c_int32),
        ("big_stage", C.c_int32), ("big_pitch", C.c_int32),
        ("onehot", C.c_int32), ("rgb16", C.c_int32), ("rules", C.c_int32),
        ("specialised", C.c_int32), ("phase_kernel", C.c_int32), ("rollout_in_one_launch", C.c_int32),
        ("walk_blocks", C.c_int32),
        ("walk_min_envs", C.c_int64), ("walk_max_envs", C.c_int64), ("big_stage_min_envs", C.c_int64),
        ("kernel", C.c_char * 192), ("kernel_prebuilt", C.c_char * 192), ("kernel_plain", C.c_char * 192),
        ("kernel_rollout", C.c_char * 192), ("kernel_walk", C.c_char * 192),
        ("kernel_phase", C.c_char * 96), ("kernel_observe_rows", C.c_char * 96),
    ]

    def as_dict(self) -> dict:
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = v.decode() if isinstance(v, bytes) else int(v)
        return out


class SgwTurnRows(C.Structure):
    """Mirror of ``struct sgw_turn_rows`` (include/sgw.h): the agents' replay rings for ``sgw_turn_bind``."""

    _fields_ = [
        ("states", C.c_void_p * MAX_AGENTS), ("rewards", C.c_void_p * MAX_AGENTS), ("actions", C.c_void_p * MAX_AGENTS),
        ("dones", C.c_void_p * MAX_AGENTS),
        ("capacity", C.c_int64 * MAX_AGENTS), ("row", C.c_int64 * MAX_AGENTS), ("step", C.c_int64 * MAX_AGENTS),
        ("row_elems", C.c_int64 * MAX_AGENTS),
    ]


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGW_LIB") or os.path.join(_HERE, "csrc", "libsgw.so")   # SGW_LIB: diagnostic builds (tools/)

# every symbol include/sgw.h declares
EXPORTS = (
    "sgw_create", "sgw_destroy", "sgw_reset", "sgw_observe", "sgw_step", "sgw_rollout", "sgw_reduce_metrics",
    "sgw_random_actions", "sgw_set_obs_format", "sgw_bind_agent_state", "sgw_init_agent_state", "sgw_bind_agent_dir", "sgw_get_status", "sgw_obs_elems_per_env", "sgw_grid_bytes_per_env",
    "sgw_algorithmic_bytes_per_env_step", "sgw_set_timing", "sgw_get_step_time_ms", "sgw_get_step_times_ms",
    "sgw_set_auto_reset", "sgw_set_wg_per_cu", "sgw_launch_info", "sgw_capabilities", "sgw_observe_rows", "sgw_act", "sgw_observe_full",
    "sgw_set_option", "sgw_plan", "sgw_jit_stats", "sgw_jit_compile", "sgw_bind_row_tail",
    "sgw_turn_bind", "sgw_turn_set", "sgw_turn_begin", "sgw_turn_act", "sgw_turn_end", "sgw_turn_state", "sgw_turn_begin_rows", "sgw_turn_act_rows", "sgw_turn_epsilon", "sgw_turn_prev_rows", "sgw_turn_resolve", "sgw_gather_rows", "sgw_sweep_observe_rows", "sgw_choose_actions", "sgw_verify_rows", "sgw_apply_actions",
    "sgw_last_error", "sgw_version",
)

_lib = None


class SgwError(RuntimeError):
    pass


def _hip_runtimes_mapped():
    paths = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    paths.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return sorted(paths)


def load():
    """Load libsgw.so (after torch, so that it binds to torch's HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise SgwError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the step/observe path."
        )
    import torch  # noqa: F401  (maps torch's libamdhip64 first; libsgw.so then resolves to the same soname)

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise SgwError(f"two HIP runtimes are mapped in this process ({rts}); stream handles would not be shared")
    vp, u8p, f32p, f64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    cfgp = C.POINTER(SgwConfig)
    lib.sgw_create.argtypes = [cfgp, C.POINTER(vp)]
    lib.sgw_create.restype = C.c_int
    lib.sgw_destroy.argtypes = [vp]
    lib.sgw_destroy.restype = None
    lib.sgw_reset.argtypes = [vp, u8p, u8p, f64p, C.c_uint32, vp]
    lib.sgw_reset.restype = C.c_int
    lib.sgw_observe.argtypes = [vp, u8p, u8p, f32p, C.c_int32, C.c_int32, vp]
    lib.sgw_observe.restype = C.c_int
    lib.sgw_step.argtypes = [vp, u8p, u8p, u8p, f32p, f32p, f64p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                             C.c_uint32, vp]
    lib.sgw_step.restype = C.c_int
    lib.sgw_rollout.argtypes = [vp, u8p, u8p, u8p, f32p, f32p, f64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64,
                                C.c_int64, C.c_uint32, vp]
    lib.sgw_rollout.restype = C.c_int
    lib.sgw_reduce_metrics.argtypes = [vp, f64p, f64p, vp]
    lib.sgw_reduce_metrics.restype = C.c_int
    lib.sgw_random_actions.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, vp]
    lib.sgw_random_actions.restype = C.c_int
    lib.sgw_set_obs_format.argtypes = [vp, C.c_int]
    lib.sgw_set_obs_format.restype = C.c_int
    lib.sgw_bind_agent_state.argtypes = [vp, u8p, u8p]
    lib.sgw_bind_agent_state.restype = C.c_int
    lib.sgw_bind_agent_dir.argtypes = [vp, u8p]
    lib.sgw_bind_agent_dir.restype = C.c_int
    lib.sgw_init_agent_state.argtypes = [vp, u8p, vp]
    lib.sgw_init_agent_state.restype = C.c_int
    lib.sgw_get_status.argtypes = [vp, C.POINTER(C.c_int32), vp]
    lib.sgw_get_status.restype = C.c_int
    for name in ("sgw_obs_elems_per_env", "sgw_grid_bytes_per_env", "sgw_algorithmic_bytes_per_env_step"):
        getattr(lib, name).argtypes = [cfgp]
        getattr(lib, name).restype = C.c_int64
    lib.sgw_set_timing.argtypes = [vp, C.c_int]
    lib.sgw_set_timing.restype = C.c_int
    lib.sgw_get_step_time_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.sgw_get_step_time_ms.restype = C.c_int
    lib.sgw_get_step_times_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]
    lib.sgw_get_step_times_ms.restype = C.c_int
    lib.sgw_set_auto_reset.argtypes = [vp, C.c_uint32, f64p]
    lib.sgw_set_auto_reset.restype = C.c_int
    lib.sgw_set_wg_per_cu.argtypes = [vp, C.c_int]
    lib.sgw_set_wg_per_cu.restype = C.c_int
    lib.sgw_launch_info.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.sgw_launch_info.restype = C.c_int
    lib.sgw_observe_full.argtypes = [vp, u8p, vp, vp]
    lib.sgw_observe_full.restype = C.c_int
    lib.sgw_capabilities.argtypes = [vp]
    lib.sgw_capabilities.restype = C.c_int
    lib.sgw_observe_rows.argtypes = [vp, u8p, u8p, C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_int32, vp]
    lib.sgw_observe_rows.restype = C.c_int
    lib.sgw_sweep_observe_rows.argtypes = [vp, u8p, u8p, C.POINTER(C.c_void_p), C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    lib.sgw_sweep_observe_rows.restype = C.c_int
    lib.sgw_act.argtypes = [vp, u8p, u8p, u8p, C.POINTER(C.c_void_p), C.c_int64, f32p, f64p, C.c_int32, vp, C.c_int32, vp, vp, vp]
    lib.sgw_act.restype = C.c_int
    lib.sgw_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.sgw_set_option.restype = C.c_int
    lib.sgw_plan.argtypes = [cfgp, C.c_int32, C.c_int64, C.POINTER(SgwPlan)]
    lib.sgw_plan.restype = C.c_int
    lib.sgw_jit_compile.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64]
    lib.sgw_jit_compile.restype = C.c_int
    lib.sgw_jit_stats.argtypes = [C.POINTER(C.c_double)]
    lib.sgw_jit_stats.restype = C.c_int
    lib.sgw_bind_row_tail.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.sgw_bind_row_tail.restype = C.c_int
    lib.sgw_turn_bind.argtypes = [vp, C.POINTER(SgwTurnRows)]
    lib.sgw_turn_bind.restype = C.c_int
    lib.sgw_turn_set.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    lib.sgw_turn_set.restype = C.c_int
    lib.sgw_turn_begin.argtypes = [vp, u8p, u8p, u8p, f32p, f32p, f64p, C.c_uint32, vp]
    lib.sgw_turn_begin.restype = C.c_int
    lib.sgw_turn_act.argtypes = [vp, u8p, u8p, u8p, vp, f32p, f64p, C.c_int32, vp, C.c_int32, vp]
    lib.sgw_turn_act.restype = C.c_int
    lib.sgw_turn_begin_rows.argtypes = [vp, u8p, u8p, u8p, f32p, f64p, C.POINTER(C.c_void_p), C.c_int64, C.c_uint32, vp]
    lib.sgw_turn_begin_rows.restype = C.c_int
    lib.sgw_turn_act_rows.argtypes = [vp, u8p, u8p, u8p, C.POINTER(C.c_void_p), C.c_int64, f32p, f64p, C.c_int32, vp, C.c_int32, vp]
    lib.sgw_turn_act_rows.restype = C.c_int
    lib.sgw_turn_end.argtypes = [vp, vp, vp]
    lib.sgw_turn_end.restype = C.c_int
    lib.sgw_turn_state.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int64), vp]
    lib.sgw_turn_state.restype = C.c_int
    lib.sgw_turn_resolve.argtypes = [vp, u8p, u8p, u8p, vp, C.c_int64, f32p, f64p, vp, vp, vp, vp, C.c_int64, vp, vp, C.c_int32, vp]
    lib.sgw_turn_resolve.restype = C.c_int
    lib.sgw_gather_rows.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp]
    lib.sgw_gather_rows.restype = C.c_int
    lib.sgw_verify_rows.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp, vp]
    lib.sgw_verify_rows.restype = C.c_int
    lib.sgw_apply_actions.argtypes = [vp, vp, vp, vp, C.c_int64, vp]
    lib.sgw_apply_actions.restype = C.c_int
    lib.sgw_choose_actions.argtypes = [vp, vp, vp, C.c_int64, C.c_uint32, C.c_uint32, vp, vp]
    lib.sgw_choose_actions.restype = C.c_int
    lib.sgw_turn_prev_rows.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    lib.sgw_turn_prev_rows.restype = C.c_int
    lib.sgw_turn_epsilon.argtypes = [vp, C.c_int32, C.c_double, vp]
    lib.sgw_turn_epsilon.restype = C.c_int
    lib.sgw_last_error.argtypes = []
    lib.sgw_last_error.restype = C.c_char_p
    lib.sgw_version.argtypes = []
    lib.sgw_version.restype = C.c_char_p
    _lib = lib
    # tools pass dispatcher options to child processes as SGW_OPTIONS="key=value,key=value": read HERE, by the Python
    # tooling -- the library itself reads no environment variable except SGW_DEBUG
    for item in filter(None, os.environ.get("SGW_OPTIONS", "").replace(";", ",").split(",")):
        key, _, value = item.partition("=")
        set_option(key.strip(), value.strip())
        _opt_baseline[key.strip()] = value.strip()
    return lib


_opt_values = {}      # process-wide options as they stand (key -> text), mirrored here so that a block can put back what it found
_opt_baseline = {}    # ... as SGW_OPTIONS set them when the library was loaded


def set_option(key, value=None, engine=None) -> None:
    """``sgw_set_option``: a dispatcher knob (``sorrel_amd/csrc/options.h``) for engines created from now on, or -- the live
    keys only -- for one engine handle.  ``value=None`` restores the key's default, ``key=None`` all of them."""
    lib = load()
    k = None if key is None else str(key).encode()
    v = None if value is None else str(int(value) if isinstance(value, bool) else value).encode()
    check(lib.sgw_set_option(engine, k, v))
    if engine is None:
        if key is None:
            _opt_values.clear()
        elif value is None:
            _opt_values.pop(str(key), None)
        else:
            _opt_values[str(key)] = v.decode()


def get_option(key):
    """The process-wide value last set for ``key`` through this module (text), or None while it stands at the library's default."""
    return _opt_values.get(str(key))


def reset_options() -> None:
    """Every process-wide option back to where the process started: the library's defaults plus what ``SGW_OPTIONS`` asked for."""
    set_option(None)
    for k, v in _opt_baseline.items():
        set_option(k, v)


class options:
    """``with options(group=16, jit=0): ...`` -- process-wide options for the engines created inside the block; on exit every key goes
    back to the value it had on entry (not to the library default: blocks nest, and what ``SGW_OPTIONS`` set survives them)."""

    def __init__(self, **kw):
        self.kw = kw
        self.before = {}

    def __enter__(self):
        self.before = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.before.items():
            set_option(k, v)
        return False


def plan(cfg: "SgwConfig", num_cus: int = 256, lds_per_workgroup: int = 160 * 1024) -> dict:
    """``sgw_plan``: kernel family, LDS layout, staging and instances ``sgw_create`` would choose -- no device needed."""
    out = SgwPlan()
    check(load().sgw_plan(C.byref(cfg), num_cus, lds_per_workgroup, C.byref(out)))
    return out.as_dict()


def jit_compile(instance: str, arch: str = "gfx950") -> str:
    """``sgw_jit_compile``: the cache file of one specialised instance, compiled now if the cache does not hold it (no device needed)."""
    buf = C.create_string_buffer(1024)
    check(load().sgw_jit_compile(instance.encode(), arch.encode(), buf, 1024))
    return buf.value.decode()


def jit_code_object(path: str):
    """(lowered kernel name, code object bytes) of a cache file."""
    with open(path, "rb") as fh:
        data = fh.read()
    assert data[:8] == b"SGWJIT2\n", path
    n = int.from_bytes(data[8:12], "little")
    return data[12:12 + n].decode(), data[12 + n + 16:]          # (u64 size + u64 checksum in front of the code object)


def jit_stats() -> dict:
    buf = (C.c_double * 6)()
    check(load().sgw_jit_stats(buf))
    return dict(compiled=int(buf[0]), disk_hits=int(buf[1]), mem_hits=int(buf[2]), failed=int(buf[3]),
                compile_ms=float(buf[4]), load_ms=float(buf[5]))


def check(rc: int) -> None:
    if rc != OK:
        msg = load().sgw_last_error().decode("utf-8", "replace")
        if rc == EINVAL:
            raise ValueError(msg)   # the reference raises ValueError/TypeError for bad specs
        raise SgwError(f"sgw error {rc}: {msg}")
