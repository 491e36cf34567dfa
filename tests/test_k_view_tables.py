"""The static tables behind k_view's laser_map, checked on the CPU (no GPU, no oracle): tests/host/k_view_tables_check.cpp
builds a robot class with the library's own host code (img_env_amd/csrc/host_tables.h) and checks
  * that the cells a beam's hit "leaves alone" (agent.cpp:555-560) are one run of steps right behind the hit (what the
    kernel's packed hit word rests on) and that `ray_run` holds its length;
  * that the per-cell ray lists, their heads, the 8-byte cell records and the reach-table blocks agree with each other;
  * that the table-driven composition the kernel performs -- the top beam's verdict, else the first deciding entry of the
    cell's list, behind the reach filter -- gives the laser_map of the reference's sequential beam-after-beam algorithm on
    random occupancies with and without axis-parallel walls;
  * that the static lists a STEP walks instead of every cell (round 4: `dyn_groups` of k_view, `tap_chunks` of k_taps_big) hold
    exactly the groups of cells / chunks of pixels a beam can reach, and that a group's word carries its field-of-view and
    own-footprint bits."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kvt") / "k_view_tables_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "k_view_tables_check.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("view_w,view_h,res,beams,a0,a1", [
    (12, 12, 0.25, 360, -1.5708, 1.5708),      # the headline geometry: 48 x 48 cells, 360 beams
    (11.25, 12.5, 0.25, 200, -3.14159, 3.14159),  # 50 x 45 cells (rows not a multiple of 4), full circle
    (6, 6, 0.125, 720, -1.5708, 1.5708),       # 48 x 48 at half the cell size, 720 beams
    (5, 7, 0.25, 33, -0.7, 2.1),               # a small odd view, few beams
    (24, 24, 0.25, 960, -1.5708, 1.5708),      # cfg-5's 96 x 96 cells, 960 beams
])
def test_tables_reproduce_the_sequential_laser_map(checker, view_w, view_h, res, beams, a0, a1):
    out = subprocess.run([checker, str(view_w), str(view_h), str(res), str(beams), str(a0), str(a1), "7"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


@pytest.fixture(scope="module")
def big_checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvt") / "big_view_tables_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "big_view_tables_check.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("view_w,view_h,res,beams,a0,a1,map_h,map_w", [
    (6, 6, 0.015, 1000, -1.5708, 1.5708, 733, 733),  # the reference's shipped test.yaml: 400 x 400 cells, 1000 beams, 733 x 733 map
    (12, 12, 0.25, 360, -1.5708, 1.5708, 400, 400),  # the headline view through the tiled kernels (IMGENV_FLAG_VIEW_TILED)
    (11.25, 12.5, 0.25, 200, -3.14159, 3.14159, 61, 203),  # 50 x 45 cells (not multiples of 8), full circle; an odd map
    (5, 7, 0.25, 33, -0.7, 2.1, 9, 4100),            # a small odd view; a map wider than 4096 cells
])
def test_big_view_tables_and_crop_map_indices(big_checker, view_w, view_h, res, beams, a0, a1, map_h, map_w):
    """tests/host/big_view_tables_check.cpp: path table (word address << 5 | bit entries) against the per-cell ray lists, crop tile
    list against the field of view, and crop_map's index arithmetic (blocked index in its six-instruction form, row of a cell
    index by multiply-shift) against their definitions"""
    out = subprocess.run([big_checker, str(view_w), str(view_h), str(res), str(beams), str(a0), str(a1), str(map_h), str(map_w)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
