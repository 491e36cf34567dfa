"""The C ABI without a GPU: the HIP library loads, exports every function include/imgenv.h declares, and the
ctypes mirror of the structs has the layout the C compiler gives them.  No compute call is made."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "imgenv.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(imgenv_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def hip_lib():
    from img_env_amd import _cabi
    path = _cabi.library_path()
    if not os.path.exists(path):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return C.CDLL(path)


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("imgenv_create", "imgenv_reset", "imgenv_reset_world", "imgenv_reset_worlds", "imgenv_reset_worlds_spawn",
                 "imgenv_spawn", "imgenv_step", "imgenv_step_autoreset", "imgenv_step_begin",
                 "imgenv_step_end", "imgenv_outputs", "imgenv_destroy", "imgenv_comm_init"):
        assert must in names


def test_library_exports_every_declared_function(hip_lib):
    missing = [f for f in declared_functions() if not hasattr(hip_lib, f)]
    assert not missing, missing


def test_python_symbol_list_matches_the_header():
    from img_env_amd import _cabi
    assert set(_cabi.SYMBOLS) <= set(declared_functions())


def test_library_identifies_itself_without_a_device(hip_lib):
    from img_env_amd import _cabi
    hip_lib.imgenv_backend.restype = C.c_char_p
    hip_lib.imgenv_abi_version.restype = C.c_int32
    assert hip_lib.imgenv_backend() == b"hip-gfx950"
    assert hip_lib.imgenv_abi_version() == _cabi.ABI_VERSION
    hip_lib.imgenv_kernel_name.restype = C.c_char_p
    names = [hip_lib.imgenv_kernel_name(i).decode() for i in range(_cabi.K_COUNT)]
    assert names[5] == "k_view" and names[6] == "k_obs" and all(names)


def test_ctypes_structs_match_the_c_layout(tmp_path):
    """sizeof / offsetof as gcc sees include/imgenv.h vs the ctypes mirror"""
    from img_env_amd import _cabi
    probe = tmp_path / "probe.c"
    fields = {"imgenv_cfg": ["abi_version", "view_resolution", "robot_shape", "limiter_w", "image_size", "ped_image_r",
                             "robot_size_last", "time_max", "out_arena", "out_arena_bytes"],
              "imgenv_reset_batch": ["n_obstacles", "obs_pose", "ped_traj", "ped_traj_cap", "ignore_obstacle", "ped_traj_v"],
              "imgenv_out": ["n_local", "vector_states", "lasers", "paper_rewards", "counters", "step_rewards", "step_is_collisions", "step_all_down"],
              "imgenv_spawn_agent": ["target_type", "begin", "target", "module_size", "begin_multi", "target_multi", "n_target_multi"],
              "imgenv_spawn_obstacle": ["pose_type", "size_range", "pose"],
              "imgenv_spawn_cfg": ["n_obstacles", "agents", "obstacles", "clearance", "target_min_dist", "circle_ranges", "go_back", "ignore_obstacle"]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "imgenv.h"', "int main(void) {"]
    for st, fs in fields.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fs:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    lines.append("return 0; }")
    probe.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)]).decode().splitlines())
    mirror = {"imgenv_cfg": _cabi.Cfg, "imgenv_reset_batch": _cabi.ResetBatch, "imgenv_out": _cabi.Out,
              "imgenv_spawn_agent": _cabi.SpawnAgent, "imgenv_spawn_obstacle": _cabi.SpawnObstacle, "imgenv_spawn_cfg": _cabi.SpawnCfg}
    for st, fs in fields.items():
        assert int(got[st]) == C.sizeof(mirror[st]), st
        for f in fs:
            assert int(got["%s.%s" % (st, f)]) == getattr(mirror[st], f).offset, (st, f)


def test_oracle_library_exports_its_entry_points(oracle_lib):
    for f in ("oracle_create", "oracle_reset", "oracle_step", "oracle_step_begin", "oracle_step_end", "oracle_records",
              "oracle_outputs", "oracle_destroy", "oracle_last_error"):
        assert hasattr(oracle_lib, f), f
