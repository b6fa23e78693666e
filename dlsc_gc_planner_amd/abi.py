"""ctypes mirror of include/dlscqp.h (the C ABI of the HIP trajectory optimizer).

Only struct layouts, constants and the loader live here.  The loader fails loudly when the HIP
library has not been built: there is no CPU fallback in the product path.
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 2

# API return codes
SUCCESS = 0
ERR_INVALID_ARGUMENT = -1
ERR_HIP = -2
ERR_NO_DEVICE = -3
ERR_UNSUPPORTED = -4
ERR_SELFTEST = -5

# per-agent verdicts
OK = 0
INFEASIBLE = 1
ITERATION_LIMIT = 2
NUMERICAL = 3
GOAL_FALLBACK = 16

# MATP::PlannerMode (sp_const.hpp:18-26)
PLANNER_DLSCGC, PLANNER_DLSC, PLANNER_LSC, PLANNER_BVC, PLANNER_ORCA, PLANNER_RECIPROCALRSFC, PLANNER_CIRCLETEST = range(7)
# MATP::SlackMode (sp_const.hpp:44-48)
SLACK_NONE, SLACK_CONTINUITY, SLACK_COLLISIONCONSTRAINT = range(3)

ROW_DYNAMIC = 1


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("world_dimension", C.c_int32),
        ("M", C.c_int32),
        ("n", C.c_int32),
        ("phi", C.c_int32),
        ("phi_n", C.c_int32),
        ("planner_mode", C.c_int32),
        ("slack_mode", C.c_int32),
        ("world_use_octomap", C.c_int32),
        ("device", C.c_int32),
        ("dt", C.c_double),
        ("control_input_weight", C.c_double),
        ("terminal_weight", C.c_double),
        ("slack_collision_weight", C.c_double),
        ("communication_range", C.c_double),
        ("world_z_2d", C.c_double),
        ("slack_threshold", C.c_double),
        ("world_min", C.c_double * 3),
        ("world_max", C.c_double * 3),
    ]


class Agent(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("velocity", C.c_float * 3),
        ("acceleration", C.c_float * 3),
        ("current_goal_point", C.c_float * 3),
        ("next_waypoint", C.c_float * 3),
        ("id", C.c_int32),
        ("n_obs", C.c_int32),
        ("n_slack_obs", C.c_int32),
        ("max_vel", C.c_double),
        ("max_acc", C.c_double),
        ("radius", C.c_double),
        ("nominal_velocity", C.c_double),
    ]


class LscRow(C.Structure):
    _fields_ = [
        ("normal", C.c_float * 3),
        ("point", C.c_float * 3),
        ("d", C.c_double),
        ("m", C.c_int16),
        ("i", C.c_int8),
        ("flags", C.c_uint8),
        ("obs", C.c_int16),
        ("slack", C.c_int16),
    ]


class DeviceBatch(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32),
        ("max_slack_obs", C.c_int32),
        ("do_goal", C.c_int32),
        ("do_traj", C.c_int32),
        ("max_rows_per_agent", C.c_int32),
        ("launch_offset", C.c_int32),
        ("launch_count", C.c_int32),
        ("reserved_", C.c_int32),
        ("agents", C.c_void_p),
        ("row_offsets", C.c_void_p),
        ("nx", C.c_void_p),
        ("ny", C.c_void_p),
        ("nz", C.c_void_p),
        ("px", C.c_void_p),
        ("py", C.c_void_p),
        ("pz", C.c_void_p),
        ("d", C.c_void_p),
        ("key", C.c_void_p),
        ("sfc_boxes", C.c_void_p),
        ("out_goal", C.c_void_p),
        ("out_goal_status", C.c_void_p),
        ("out_ctrl", C.c_void_p),
        ("out_ctrl_f64", C.c_void_p),
        ("out_cost", C.c_void_p),
        ("out_slack_cost", C.c_void_p),
        ("out_status", C.c_void_p),
        ("out_iters", C.c_void_p),
        ("agent_order", C.c_void_p),
        ("debug_stamps", C.c_void_p),
        ("row_counts", C.c_void_p),
        ("row_pitch", C.c_int32),
        ("hint_steps", C.c_int32),
        ("hint_iters", C.c_void_p * 3),
        ("out_record_stride", C.c_int32),
        ("reserved2_", C.c_int32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("hint_mask_in", C.c_void_p),
        ("hint_mask_out", C.c_void_p),
        ("hint_mask_zero", C.c_void_p),
        ("advance_after", C.c_void_p),
        ("out_conflict", C.c_void_p),      # round 6: the certificates of INFEASIBLE verdicts ([n_agents][conflict_stride] doubles)
        ("conflict_stride", C.c_int32),
        ("reserved3_", C.c_int32),
    ]


class LscGen(C.Structure):
    _fields_ = [
        ("n_local", C.c_int32),
        ("first_plan", C.c_int32),
        ("local_ids", C.c_void_p),
        ("plans", C.c_void_p),
        ("position", C.c_void_p),
        ("velocity", C.c_void_p),
        ("goals", C.c_void_p),
        ("radius", C.c_void_p),
        ("downwash", C.c_void_p),
        ("nbr_offsets", C.c_void_p),
        ("nbr_ids", C.c_void_p),
        ("nx", C.c_void_p), ("ny", C.c_void_p), ("nz", C.c_void_p),
        ("px", C.c_void_p), ("py", C.c_void_p), ("pz", C.c_void_p),
        ("d", C.c_void_p),
        ("key", C.c_void_p),
        ("n_dyn", C.c_int32),
        ("dyn_size_prediction", C.c_int32),
        ("dyn_uncertainty_horizon", C.c_double),
        ("dyn_position", C.c_void_p),
        ("dyn_velocity", C.c_void_p),
        ("dyn_radius", C.c_void_p),
        ("dyn_max_acc", C.c_void_p),
        ("dyn_downwash", C.c_void_p),
        ("observed_position", C.c_void_p),
        ("reset_threshold", C.c_double),
        ("prune", C.c_int32),
        ("reserved_", C.c_int32),
        ("prune_sfc_boxes", C.c_void_p),
        ("out_row_counts", C.c_void_p),
        ("nbr_counts", C.c_void_p),
        ("tail_from", C.c_int32),
        ("tail_segments", C.c_int32),
        ("max_neighbours", C.c_int32),
        ("reserved2_", C.c_int32),
    ]


class NeighbourSearch(C.Structure):  # dlscqp_neighbour_search
    _fields_ = [
        ("n_total", C.c_int32),
        ("n_local", C.c_int32),
        ("local_ids", C.c_void_p),
        ("position", C.c_void_p),
        ("position_stride", C.c_int32),
        ("pitch", C.c_int32),
        ("out_nbr_ids", C.c_void_p),
        ("out_nbr_counts", C.c_void_p),
        ("out_row_counts", C.c_void_p),
        ("n_dyn", C.c_int32),
        ("reserved_", C.c_int32),
        ("out_overflow", C.c_void_p),
    ]


class NeighbourKnn(C.Structure):  # dlscqp_neighbour_knn
    _fields_ = [("n_total", C.c_int32), ("group", C.c_int32), ("k", C.c_int32), ("pitch", C.c_int32), ("position", C.c_void_p),
                ("position_stride", C.c_int32), ("reserved_", C.c_int32), ("out_nbr_ids", C.c_void_p), ("out_nbr_counts", C.c_void_p)]


class WaypointTrap(C.Structure):  # dlscqp_waypoint_trap
    _fields_ = [("agents", C.c_void_p), ("sfc_boxes", C.c_void_p), ("out_released", C.c_void_p)]


class SfcGen(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32),
        ("initialize", C.c_int32),
        ("grid_resolution", C.c_double),
        ("radius", C.c_void_p),
        ("max_vel", C.c_void_p),
        ("position", C.c_void_p),
        ("initial_traj", C.c_void_p),
        ("goals", C.c_void_p),
        ("waypoints", C.c_void_p),
        ("sfcs", C.c_void_p),
        ("out_code", C.c_void_p),
    ]


class Advance(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32),
        ("status", C.c_void_p),
        ("goal_status", C.c_void_p),
        ("ctrl", C.c_void_p),
        ("goal", C.c_void_p),
        ("plans", C.c_void_p),
        ("current_goal", C.c_void_p),
        ("initial_traj", C.c_void_p),
        ("agents", C.c_void_p),
        ("waypoints", C.c_void_p),
        ("desired_goal", C.c_void_p),
        ("waypoint_step", C.c_double),
        ("waypoint_hold_on_lp_failure", C.c_int32),
        ("reserved_", C.c_int32),
        ("out_boxes", C.c_void_p),
        ("box_margin", C.c_float),
        ("box_min", C.c_float * 3),
        ("box_max", C.c_float * 3),
        ("reserved2_", C.c_int32),
        ("out_host_pack", C.c_void_p),
        ("alert_slack_cost", C.c_void_p),
        ("alert_slack_stride", C.c_int32),
        ("n_alert_obs", C.c_int32),
        ("alert_threshold", C.c_double),
    ]


class ExchangeConfig(C.Structure):  # dlscqp_exchange_config
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("n_local", C.c_int32), ("local_only", C.c_int32), ("unique_id", C.c_void_p),
                ("recv_buffers", C.c_void_p * 2)]


class StepGroup(C.Structure):  # dlscqp_step_group
    _fields_ = [("ctx", C.c_void_p), ("batch", C.POINTER(DeviceBatch)), ("stream", C.c_void_p)]


class StepArgs(C.Structure):  # dlscqp_step_args
    _fields_ = [("n_groups", C.c_int32), ("exchange_mode", C.c_int32), ("groups", C.POINTER(StepGroup)), ("exchange", C.c_void_p),
                ("ctrl", C.c_void_p), ("goal", C.c_void_p), ("status", C.c_void_p), ("records", C.c_void_p),
                ("rotation", C.c_int32), ("reserved_", C.c_int32)]


EXCHANGE_NONE, EXCHANGE_PIPELINED, EXCHANGE_SERIAL = 0, 1, 2
EXCHANGE_ID_BYTES = 128

assert C.sizeof(Config) == 144, C.sizeof(Config)
assert C.sizeof(Agent) == 104, C.sizeof(Agent)
assert C.sizeof(LscRow) == 40, C.sizeof(LscRow)

AGENT_DTYPE = np.dtype(Agent)
ROW_DTYPE = np.dtype(LscRow)


def make_config(dim=3, M=5, n=5, phi=3, phi_n=1, dt=0.2, control_input_weight=0.01,
                terminal_weight=1.0, slack_collision_weight=1.0, communication_range=-1.0,
                world_use_octomap=False, world_min=(-5.0, -5.0, 0.0), world_max=(5.0, 5.0, 2.5),
                world_z_2d=1.0, slack_threshold=0.1, planner_mode=PLANNER_DLSCGC,
                slack_mode=SLACK_NONE, device=0):
    """Config with the values of launch/testall_DLSCGC_empty.launch unless overridden.

    world_min/world_max are rounded through float32 like Mission's point3d (mission.hpp:31)."""
    c = Config()
    c.abi_version = ABI_VERSION
    c.world_dimension = dim
    c.M, c.n, c.phi, c.phi_n = M, n, phi, phi_n
    c.planner_mode = planner_mode
    c.slack_mode = slack_mode
    c.world_use_octomap = 1 if world_use_octomap else 0
    c.device = device
    c.dt = dt
    c.control_input_weight = control_input_weight
    c.terminal_weight = terminal_weight
    c.slack_collision_weight = slack_collision_weight
    c.communication_range = communication_range
    c.world_z_2d = world_z_2d
    c.slack_threshold = slack_threshold
    for k in range(3):
        c.world_min[k] = float(np.float32(world_min[k]))
        c.world_max[k] = float(np.float32(world_max[k]))
    return c


def key_of(m, i, slack, dyn):
    """DLSCQP_KEY of include/dlscqp.h (vectorised)."""
    m = np.asarray(m).astype(np.uint32)
    i = np.asarray(i).astype(np.uint32)
    slack = np.asarray(slack).astype(np.int64)
    dyn = np.asarray(dyn).astype(np.uint32)
    s = np.where(slack < 0, 0xFFFF, slack).astype(np.uint32)
    return (m & 0xFF) | ((i & 0xF) << 8) | ((dyn & 1) << 12) | (s << 16)


class MapfConfig(C.Structure):  # dlscqp_mapf_config
    _fields_ = [("world_dimension", C.c_int32), ("M", C.c_int32), ("world_use_octomap", C.c_int32), ("reserved_", C.c_int32),
                ("dt", C.c_double), ("grid_resolution", C.c_double), ("world_resolution", C.c_double), ("world_z_2d", C.c_double),
                ("communication_range", C.c_double), ("agent_radius", C.c_double), ("agent_downwash", C.c_double),
                ("obs_uncertainty_horizon", C.c_double), ("world_min", C.c_double * 3), ("world_max", C.c_double * 3)]


class MapfStepArgs(C.Structure):  # dlscqp_mapf_step_args
    _fields_ = [("n_agents", C.c_int32), ("n1", C.c_int32), ("position", C.c_void_p), ("start_point", C.c_void_p),
                ("desired_goal", C.c_void_p), ("current_goal", C.c_void_p), ("next_waypoint", C.c_void_p), ("plans", C.c_void_p),
                ("n_obs", C.c_int32), ("reserved_", C.c_int32), ("obs_position", C.c_void_p), ("obs_velocity", C.c_void_p),
                ("obs_radius", C.c_void_p), ("obs_max_acc", C.c_void_p), ("alert", C.c_void_p), ("out_desired", C.c_void_p),
                ("out_group", C.c_void_p)]


_LIB = None


def library_path():
    here = os.path.dirname(os.path.abspath(__file__))
    # DLSCQP_LIB selects another build of the same HIP library (the -DDLSCQP_STAMPS diagnostic build)
    return os.environ.get("DLSCQP_LIB") or os.path.join(here, "csrc", "libdlscqp.so")


def load_library():
    """Load the HIP library; raise if it has not been built (no CPU fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "dlsc_gc_planner_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    P = C.POINTER
    lib.dlscqp_create.argtypes = [P(Config), P(C.c_void_p)]
    lib.dlscqp_create.restype = C.c_int
    lib.dlscqp_destroy.argtypes = [C.c_void_p]
    lib.dlscqp_destroy.restype = None
    lib.dlscqp_update_param.argtypes = [C.c_void_p, P(Config)]
    lib.dlscqp_update_param.restype = C.c_int
    lib.dlscqp_last_error.argtypes = []
    lib.dlscqp_last_error.restype = C.c_char_p
    lib.dlscqp_lds_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.dlscqp_lds_bytes.restype = C.c_int64
    if hasattr(lib, "dlscqp_fused_step_device"):  # (round 6)
        lib.dlscqp_fused_step_device.argtypes = [C.c_void_p] * 8
        lib.dlscqp_fused_step_device.restype = C.c_int
    if hasattr(lib, "dlscqp_conflict_doubles"):  # (round 6; tests/tools/ab.py also loads libraries built from earlier rounds' sources)
        lib.dlscqp_conflict_doubles.argtypes = [C.c_void_p, C.c_int32]
        lib.dlscqp_conflict_doubles.restype = C.c_int32
        lib.dlscqp_write_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_char_p]
        lib.dlscqp_write_lp.restype = C.c_int
        lib.dlscqp_conflict_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.POINTER(ConflictMember), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        lib.dlscqp_conflict_decode.restype = C.c_int
    if hasattr(lib, "dlscqp_workspace_bytes"):  # (tests/tools/ab.py also loads libraries built from earlier rounds' sources)
        lib.dlscqp_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        lib.dlscqp_workspace_bytes.restype = C.c_int64
    vp = C.c_void_p
    lib.dlscqp_solve_batch.argtypes = [vp, C.c_int32, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp]
    lib.dlscqp_solve_batch.restype = C.c_int
    lib.dlscqp_goal_batch.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    lib.dlscqp_goal_batch.restype = C.c_int
    lib.dlscqp_plan_batch.argtypes = [vp, C.c_int32, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    lib.dlscqp_plan_batch.restype = C.c_int
    lib.dlscqp_plan_batch_device.argtypes = [vp, P(DeviceBatch), vp]
    lib.dlscqp_plan_batch_device.restype = C.c_int
    lib.dlscqp_generate_lsc_device.argtypes = [vp, P(LscGen), vp]
    lib.dlscqp_generate_lsc_device.restype = C.c_int
    if hasattr(lib, "dlscqp_selftest"):  # (tests/tools/kbench.py also loads builds of earlier rounds for A/B timing)
        lib.dlscqp_selftest.argtypes = [vp]
        lib.dlscqp_selftest.restype = C.c_int
        lib.dlscqp_neighbours_device.argtypes = [vp, P(NeighbourSearch), vp]
        lib.dlscqp_neighbours_device.restype = C.c_int
    lib.dlscqp_waypoint_trap_device.argtypes = [vp, P(LscGen), P(WaypointTrap), vp]
    lib.dlscqp_waypoint_trap_device.restype = C.c_int
    lib.dlscqp_advance_device.argtypes = [vp, P(Advance), vp]
    lib.dlscqp_advance_device.restype = C.c_int
    lib.dlscqp_rows_to_soa.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dlscqp_rows_to_soa.restype = None
    lib.dlscqp_set_timing.argtypes = [vp, C.c_int32]
    lib.dlscqp_set_timing.restype = C.c_int
    lib.dlscqp_get_kernel_times.argtypes = [vp, vp, C.c_int32, P(C.c_int32)]
    lib.dlscqp_get_kernel_times.restype = C.c_int
    lib.dlscqp_mapf_create.argtypes = [P(MapfConfig), vp, C.c_int32, P(vp)]
    lib.dlscqp_mapf_create.restype = C.c_int
    lib.dlscqp_mapf_destroy.argtypes = [vp]
    lib.dlscqp_mapf_destroy.restype = None
    lib.dlscqp_mapf_step.argtypes = [vp, P(MapfStepArgs)]
    lib.dlscqp_mapf_step.restype = C.c_int
    lib.dlscqp_launch_order_device.argtypes = [vp, vp, C.c_int32, vp, vp]
    lib.dlscqp_launch_order_device.restype = C.c_int
    lib.dlscqp_launch_order_history_device.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp]
    lib.dlscqp_launch_order_history_device.restype = C.c_int
    if hasattr(lib, "dlscqp_neighbours_knn_device"):
        lib.dlscqp_neighbours_knn_device.argtypes = [vp, P(NeighbourKnn), vp]
        lib.dlscqp_neighbours_knn_device.restype = C.c_int
    if hasattr(lib, "dlscqp_exchange_create"):
        lib.dlscqp_exchange_unique_id.argtypes = [vp]
        lib.dlscqp_exchange_unique_id.restype = C.c_int
        lib.dlscqp_exchange_create.argtypes = [vp, P(ExchangeConfig), P(vp)]
        lib.dlscqp_exchange_create.restype = C.c_int
        lib.dlscqp_exchange_destroy.argtypes = [vp]
        lib.dlscqp_exchange_destroy.restype = None
        lib.dlscqp_exchange_post.argtypes = [vp, vp, vp, vp, vp]
        lib.dlscqp_exchange_post.restype = C.c_int
        lib.dlscqp_exchange_post_records.argtypes = [vp, vp, vp]
        lib.dlscqp_exchange_post_records.restype = C.c_int
        lib.dlscqp_exchange_before_overwrite.argtypes = [vp, vp]
        lib.dlscqp_exchange_before_overwrite.restype = C.c_int
        lib.dlscqp_exchange_before_overwrite_rotating.argtypes = [vp, vp, C.c_int32]
        lib.dlscqp_exchange_before_overwrite_rotating.restype = C.c_int
        if hasattr(lib, "dlscqp_exchange_wait_packed"):
            lib.dlscqp_exchange_wait_packed.argtypes = [vp, vp]
            lib.dlscqp_exchange_wait_packed.restype = C.c_int
        lib.dlscqp_exchange_wait.argtypes = [vp, vp, P(vp)]
        lib.dlscqp_exchange_wait.restype = C.c_int
        lib.dlscqp_exchange_set_timing.argtypes = [vp, C.c_int32]
        lib.dlscqp_exchange_set_timing.restype = C.c_int
        lib.dlscqp_exchange_get_times.argtypes = [vp, vp, vp, vp, C.c_int32, P(C.c_int32)]
        lib.dlscqp_exchange_get_times.restype = C.c_int
        lib.dlscqp_step_device.argtypes = [P(StepArgs)]
        lib.dlscqp_step_device.restype = C.c_int
    if hasattr(lib, "dlscqp_mask_order_preview_device"):
        lib.dlscqp_mask_order_preview_device.argtypes = [vp, vp, C.c_int32, vp, vp]
        lib.dlscqp_mask_order_preview_device.restype = C.c_int
    if hasattr(lib, "dlscqp_hint_order_preview_device"):
        lib.dlscqp_hint_order_preview_device.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp]
        lib.dlscqp_hint_order_preview_device.restype = C.c_int
    _LIB = lib
    return lib


class ConflictMember(C.Structure):  # dlscqp_conflict_member
    _fields_ = [("ref_index", C.c_int32), ("kind", C.c_int32), ("n", C.c_int32), ("var", C.c_int32 * 4), ("coef", C.c_double * 4),
                ("lo", C.c_double), ("hi", C.c_double), ("lam", C.c_double)]


MODEL_KINDS = ("EQ", "SFC", "LSC", "VEL", "ACC", "COMM", "WAYPOINT", "VAR_LB", "VAR_UB")  # DLSCQP_MODEL_*

EXPORTED_SYMBOLS = [
    "dlscqp_create", "dlscqp_destroy", "dlscqp_update_param", "dlscqp_last_error",
    "dlscqp_lds_bytes", "dlscqp_solve_batch", "dlscqp_goal_batch", "dlscqp_plan_batch",
    "dlscqp_plan_batch_device", "dlscqp_generate_lsc_device", "dlscqp_rows_to_soa", "dlscqp_set_timing",
    "dlscqp_get_kernel_times", "dlscqp_world_create", "dlscqp_world_destroy", "dlscqp_world_prepare", "dlscqp_generate_sfc_device",
    "dlscqp_advance_device", "dlscqp_waypoint_trap_device", "dlscqp_mapf_create", "dlscqp_mapf_destroy", "dlscqp_mapf_step",
    "dlscqp_launch_order_device", "dlscqp_launch_order_history_device", "dlscqp_neighbours_device", "dlscqp_selftest",
    "dlscqp_hint_order_preview_device",
    "dlscqp_exchange_unique_id", "dlscqp_exchange_create", "dlscqp_exchange_destroy", "dlscqp_exchange_post", "dlscqp_exchange_post_records",
    "dlscqp_exchange_before_overwrite", "dlscqp_exchange_wait", "dlscqp_exchange_set_timing", "dlscqp_exchange_get_times",
    "dlscqp_step_device", "dlscqp_neighbours_knn_device", "dlscqp_workspace_bytes", "dlscqp_mask_order_preview_device", "dlscqp_exchange_wait_packed",
    "dlscqp_conflict_doubles", "dlscqp_write_lp", "dlscqp_conflict_decode", "dlscqp_solve_batch_conflict", "dlscqp_fused_step_device",
    "dlscqp_exchange_before_overwrite_rotating",
]
