"""CPU tier: the C-ABI library loads and exports every symbol include/dlscqp.h declares, and the
ctypes mirrors have the C layouts.  No compute calls are made without a GPU."""
import ctypes as C
import os
import re
import subprocess
import tempfile

from dlsc_gc_planner_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dlscqp.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dlscqp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    fns = declared_functions()
    assert len(fns) >= 10
    lib = abi.load_library()  # raises if the HIP library was not built
    for f in fns:
        assert hasattr(lib, f), f
    assert sorted(abi.EXPORTED_SYMBOLS) == fns


def test_struct_layouts_match_the_header():
    code = r'''
#include <stdio.h>
#include <stddef.h>
#include "dlscqp.h"
int main(void) {
  printf("%zu %zu %zu %zu ", sizeof(dlscqp_config), sizeof(dlscqp_agent), sizeof(dlscqp_lsc_row), sizeof(dlscqp_device_batch));
  printf("%zu %zu %zu %zu ", offsetof(dlscqp_config, dt), offsetof(dlscqp_config, world_min), offsetof(dlscqp_agent, max_vel), offsetof(dlscqp_lsc_row, d));
  printf("%zu %zu %zu\n", offsetof(dlscqp_lsc_row, slack), offsetof(dlscqp_device_batch, agents), offsetof(dlscqp_device_batch, out_iters));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write(code)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    exp = [C.sizeof(abi.Config), C.sizeof(abi.Agent), C.sizeof(abi.LscRow), C.sizeof(abi.DeviceBatch),
           abi.Config.dt.offset, abi.Config.world_min.offset, abi.Agent.max_vel.offset, abi.LscRow.d.offset,
           abi.LscRow.slack.offset, abi.DeviceBatch.agents.offset, abi.DeviceBatch.out_iters.offset]
    assert got == exp


def test_every_struct_of_the_header_has_the_mirror_s_size_and_last_field():
    """Every structure that crosses the boundary, not only the first four: the size and the offset of the LAST field of each (the
    structures grow by appended fields; a mirror that lags behind reads garbage there -- ADVICE, round 4)."""
    pairs = [("dlscqp_config", abi.Config), ("dlscqp_agent", abi.Agent), ("dlscqp_lsc_row", abi.LscRow), ("dlscqp_device_batch", abi.DeviceBatch),
             ("dlscqp_lsc_gen", abi.LscGen), ("dlscqp_neighbour_search", abi.NeighbourSearch), ("dlscqp_neighbour_knn", abi.NeighbourKnn),
             ("dlscqp_waypoint_trap", abi.WaypointTrap), ("dlscqp_sfc_gen", abi.SfcGen), ("dlscqp_advance", abi.Advance),
             ("dlscqp_exchange_config", abi.ExchangeConfig), ("dlscqp_step_group", abi.StepGroup), ("dlscqp_step_args", abi.StepArgs),
             ("dlscqp_mapf_config", abi.MapfConfig), ("dlscqp_mapf_step_args", abi.MapfStepArgs)]
    body = "".join('  printf("%%zu %%zu ", sizeof(%s), offsetof(%s, %s));\n' % (c, c, m._fields_[-1][0]) for c, m in pairs)
    code = "#include <stdio.h>\n#include <stddef.h>\n#include \"dlscqp.h\"\nint main(void) {\n" + body + "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write(code)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    for k, (c, m) in enumerate(pairs):
        assert got[2 * k] == C.sizeof(m), (c, got[2 * k], C.sizeof(m))
        assert got[2 * k + 1] == getattr(m, m._fields_[-1][0]).offset, (c, m._fields_[-1][0])


def test_abi_version_is_checked():
    assert abi.ABI_VERSION == 2  # round 5: dlscqp_device_batch grew (workspace, hint_mask_*): a caller built against version 1 is refused
    hdr = open(os.path.join(ROOT, "include", "dlscqp.h")).read()
    assert "#define DLSCQP_ABI_VERSION 2" in hdr


def test_key_packing_matches_macro():
    import numpy as np
    assert int(abi.key_of(4, 5, -1, 0)) == (4 | (5 << 8) | (0xFFFF << 16))
    assert int(abi.key_of(9, 3, 2, 1)) == (9 | (3 << 8) | (1 << 12) | (2 << 16))
    # the library's own AoS -> SoA helper agrees (pure host code, no GPU needed)
    lib = abi.load_library()
    rows = np.zeros(3, abi.ROW_DTYPE)
    rows["m"] = [0, 4, 9]
    rows["i"] = [3, 5, 1]
    rows["slack"] = [-1, 0, 7]
    rows["flags"] = [0, 1, 1]
    rows["normal"][:, 0] = [1, 2, 3]
    rows["d"] = [0.5, 1.5, 2.5]
    out = [np.zeros(3, np.float32) for _ in range(6)] + [np.zeros(3), np.zeros(3, np.uint32)]
    lib.dlscqp_rows_to_soa(rows.ctypes.data_as(C.c_void_p), 3, *[o.ctypes.data_as(C.c_void_p) for o in out])
    assert np.array_equal(out[7], abi.key_of(rows["m"], rows["i"], rows["slack"], rows["flags"] & 1))
    assert np.array_equal(out[0], [1, 2, 3]) and np.array_equal(out[6], [0.5, 1.5, 2.5])


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        return  # covered by the gpu tier
    lib = abi.load_library()
    cfg = abi.make_config()
    h = C.c_void_p()
    rc = lib.dlscqp_create(C.byref(cfg), C.byref(h))
    assert rc == abi.ERR_NO_DEVICE
    assert b"no CPU path" in lib.dlscqp_last_error()
