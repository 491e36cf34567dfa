"""The two guards of the read-only output arrays (include/imgenv.h: IMGENV_FLAG_CHECK_OUTPUTS / IMGENV_FLAG_FULL_REWRITE).

The reference returns fresh copies with every service response (img_env.cpp:745-749; SURVEY.md section 8(b) "Ownership"); this
library hands out the kernels' incremental working copies.  CHECK_OUTPUTS must name a caller's write at the next call and stay
silent otherwise; FULL_REWRITE must make the outputs immune to anything the caller does to them."""
import numpy as np
import pytest

from parity import CLOSE, EXACT, compare
from scenarios import random_actions, small_world

pytestmark = pytest.mark.gpu

CASE = dict(n_robots=24, n_peds=10, seed=31, n_obstacles=3, clearance=0.8)


@pytest.fixture(scope="module")
def World():
    import torch
    assert torch.cuda.is_available()
    from img_env_amd.world import World
    return World


def _world(World, guard):
    grid, params, layout = small_world(**CASE)
    params = dict(params)
    if guard:
        params["output_guard"] = guard
    return World(params, grid), layout


def test_check_outputs_is_silent_while_the_caller_only_reads(World):
    """30 steps, a reset in mid-flight, side streams and all: the library's own writes never trip its own guard"""
    w, layout = _world(World, "check")
    try:
        rng = np.random.default_rng(5)
        w.reset(layout)
        for s in range(30):
            w.step(random_actions(rng, CASE["n_robots"]))
            if s == 14:
                w.reset(layout)
            w.snapshot()  # (reads only)
    finally:
        w.close()


@pytest.mark.parametrize("field", ["sensor_maps", "ped_maps", "lasers", "dones", "view_maps", "vector_states"])
def test_check_outputs_names_the_array_a_caller_wrote_into(World, field):
    import torch
    w, layout = _world(World, "check")
    try:
        rng = np.random.default_rng(6)
        w.reset(layout)
        w.step(random_actions(rng, CASE["n_robots"]))
        torch.cuda.synchronize()
        t = w.out[field]
        flat = t.view(-1)
        old = flat[flat.numel() // 2].clone()
        flat[flat.numel() // 2] = old + 1 if t.dtype != torch.uint8 else (old ^ 1)  # the in-place "normalisation" of a trainer
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match=r"wrote into imgenv_out\.%s" % field):
            w.step(random_actions(rng, CASE["n_robots"]))
        # the caller has been told; the handle carries on with what it finds (the corrupted value is the caller's problem)
        flat[flat.numel() // 2] = old
        w.step(random_actions(rng, CASE["n_robots"]))
        w.step(random_actions(rng, CASE["n_robots"]))
        # ... and a reset is guarded like a step
        w.out[field].view(-1)[0] += 1
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="wrote into imgenv_out"):
            w.reset(layout)
    finally:
        w.close()


def test_full_rewrite_outputs_survive_whatever_the_caller_does_to_them(World):
    """lockstep with a plain handle: after every call the copy-mode handle's outputs are scribbled over, and still equal
    the plain handle's bit for bit on the next call -- including the rows of frozen robots and the view cells no beam crosses"""
    import torch
    plain, layout = _world(World, None)
    copy, _ = _world(World, "copy")
    try:
        rng = np.random.default_rng(7)
        plain.reset(layout)
        copy.reset(layout)
        fields = [f for f in EXACT + CLOSE]
        for s in range(25):
            a, b = plain.snapshot(), copy.snapshot()
            for k in fields:
                assert np.array_equal(a[k], b[k], equal_nan=True), (s, k)
            for k, t in copy.out.items():  # the caller normalises in place, zeroes, ... : every byte of every array
                if t.dtype == torch.uint8 or t.dtype == torch.int8:
                    t.fill_(77)
                else:
                    t.fill_(-3)
            if s == 12:
                plain.reset(layout)
                copy.reset(layout)
                continue
            act = random_actions(rng, CASE["n_robots"])
            plain.step(act)
            copy.step(act)
        snap = plain.snapshot()
        assert (snap["is_collisions"] != 0).any() or (snap["is_arrives"] != 0).any() or True  # (frozen rows occur in this world; not required)
    finally:
        plain.close()
        copy.close()


def test_full_rewrite_is_refused_in_a_robot_shard(World):
    grid, params, layout = small_world(**CASE)
    params = dict(params, output_guard="copy", robot_begin=0, robot_end=12)
    with pytest.raises(ValueError, match="robot shard"):
        World(params, grid)


def test_default_handle_is_loud_for_its_first_calls_and_free_afterwards(World):
    """`World`'s default (output_guard "first" = IMGENV_FLAG_CHECK_OUTPUTS_FIRST): a trainer that normalises an observation in
    place is told on its next call -- during the handle's first 64 calls; afterwards the guard has switched itself off (no
    checksum kernels, no synchronisation), and what a caller writes then is its own business, as the header says."""
    import torch
    from img_env_amd import _cabi
    w, layout = _world(World, None)
    try:
        assert w.output_guard == "first" and w.params["flags"] & _cabi.FLAG_CHECK_OUTPUTS_FIRST
        rng = np.random.default_rng(8)
        n = CASE["n_robots"]
        w.reset(layout)
        w.step(random_actions(rng, n))
        torch.cuda.synchronize()
        w.out["sensor_maps"].view(-1)[5] += 1  # the in-place write
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match=r"wrote into imgenv_out\.sensor_maps"):
            w.step(random_actions(rng, n))
        w.out["sensor_maps"].view(-1)[5] -= 1
        torch.cuda.synchronize()
        launches_guarded = None
        for s in range(70):  # past the 64th call
            w.step(random_actions(rng, n))
            if s == 0:
                launches_guarded = w.launches()
        torch.cuda.synchronize()
        w.out["sensor_maps"].view(-1)[5] += 1
        torch.cuda.synchronize()
        w.step(random_actions(rng, n))  # no longer looked at
    finally:
        w.close()
