"""Stream ordering of a step's side-stream work against the CALLER's stream.

The step's observation kernel runs on a side stream beside the move and rewrites output arrays; a social-force crowd is stepped
a step ahead on a stream of its own.  Both must stay ordered behind whatever the caller has queued on its stream -- writers of the
actions, readers of the last step's outputs, the library's own FULL_REWRITE copy -- also when the host runs many steps ahead of
the device.  The parity suites synchronise after every step and cannot see such races; these tests never do."""
import numpy as np
import pytest

from parity import CLOSE, EXACT
from scenarios import random_actions, small_world

pytestmark = pytest.mark.gpu

FIELDS = tuple(k for k in EXACT + CLOSE if k != "counters")


def _delay(big, n=3):
    """a few milliseconds of work on the current stream"""
    y = big
    for _ in range(n):
        y = (y @ big) * 1e-3
    return y


def _cfg4(n):
    from test_gpu_parity import _cfg4_world
    grid, params, layout, _ = _cfg4_world(n)
    return grid, params, layout


@pytest.mark.parametrize("guard", [None, "copy"], ids=["working_copies", "full_rewrite"])
@pytest.mark.parametrize("kind,n", [("orca", 8192), ("orca", 2048), ("sfm", 8192), ("sfm", 1024)])
def test_host_running_ahead_of_the_device_changes_nothing(kind, n, guard):
    """Two handles on the same inputs.  One is stepped with a synchronisation after every step (the parity suites' regime).  The
    other gets a few milliseconds of unrelated work queued on the stream IN FRONT of every step, a stream-ordered copy of every
    output BEHIND it, the promise that its actions are complete (IMGENV_STEP_ACTIONS_READY) -- and is never waited for until
    all steps are queued.  Each step's copies must equal what the synchronised handle held after that step:
      * the early observation kernel of step t + 1 must not rewrite outputs before the caller's copy of step t's has run (rounds
        4-5: it only waited for step t's views);
      * with IMGENV_FLAG_FULL_REWRITE the library's own copy of step t must not pick up step t + 1's pedestrian data;
      * a social-force crowd computed a step ahead must not overwrite a set that has not been published yet."""
    import torch
    from img_env_amd.world import World
    if kind == "orca":
        grid, params, layout = small_world(n, 60, seed=93, grid_size=400, res=0.25, clearance=0.6, n_obstacles=2)
    else:
        grid, params, layout = _cfg4(n)
    # (World's default guard synchronises during a handle's first 64 calls: these handles must never be waited for)
    params = dict(params, output_guard=guard or "none")
    steps = 12 if n > 4096 else 24  # (every step's outputs are kept: ~0.3 GB per step at 8192 robots)
    a, b = World(dict(params), grid), World(dict(params), grid)
    try:
        a.reset(layout)
        b.reset(layout)
        torch.cuda.synchronize()
        rng = np.random.default_rng(23)
        dev = a.device
        big = torch.randn(3072, 3072, device=dev)
        acts = [torch.as_tensor(random_actions(rng, n), device=dev) for _ in range(steps)]
        want = []
        for s in range(steps):
            b.step(acts[s])
            want.append(b.snapshot())  # synchronises
        torch.cuda.synchronize()
        got = []
        keep = []
        for s in range(steps):
            keep.append(_delay(big))          # the stream is busy when the step is queued ...
            a.step(acts[s], actions_ready=True)
            got.append({k: a.out[k].clone() for k in FIELDS})  # ... and its outputs are read by stream-ordered work only
        torch.cuda.synchronize()
        for s in range(steps):
            for k in FIELDS:
                g = got[s][k].cpu().numpy()
                assert np.array_equal(g, want[s][k], equal_nan=True), (s, k)
    finally:
        a.close()
        b.close()


def test_gate_holds_while_another_stream_keeps_the_chip_busy():
    """The gate in front of the early observation (k_gate) polls a word the step's first kernel stores: it needs that kernel to
    be dispatched while the gate occupies a wavefront.  A trainer's own kernels on OTHER streams of the process compete for the
    same queues and compute units: a second stream floods the chip with long element-wise kernels while the headline-shaped
    world steps.  No step may stall (median below 5 ms, none above 50 ms -- the gate's give-up bound is 60 s) and every
    output must equal an undisturbed handle's."""
    import time
    import torch
    from img_env_amd.world import World
    n = 8192
    grid, params, layout = small_world(n, 60, seed=95, grid_size=400, res=0.25, clearance=0.6, n_obstacles=2)
    params = dict(params, output_guard="none")
    a, b = World(dict(params), grid), World(dict(params), grid)
    try:
        a.reset(layout)
        b.reset(layout)
        rng = np.random.default_rng(29)
        dev = a.device
        acts = [torch.as_tensor(random_actions(rng, n), device=dev) for _ in range(40)]
        want = []
        for s in range(40):
            b.step(acts[s])
            want.append(b.snapshot())
        noise = torch.cuda.Stream(device=dev)
        x = torch.ones(256 * 1024 * 1024 // 4, device=dev)  # 256 MiB: ~0.1 ms per pass
        times = []
        for s in range(40):
            with torch.cuda.stream(noise):
                for _ in range(40):
                    x.mul_(1.0000001)
            t0 = time.perf_counter()
            a.step(acts[s])
            torch.cuda.current_stream(dev).synchronize()
            times.append(time.perf_counter() - t0)
            got = {k: a.out[k].cpu().numpy() for k in FIELDS}
            for k in FIELDS:
                assert np.array_equal(got[k], want[s][k], equal_nan=True), (s, k)
        torch.cuda.synchronize()
        print("step latency beside a busy stream: median %.2f ms, max %.2f ms" % (1e3 * float(np.median(times)), 1e3 * max(times)))
        assert np.median(times) < 5e-3 and max(times) < 50e-3, (float(np.median(times)), max(times))
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("streams", ["one_stream", "two_streams"])
def test_two_handles_stepping_alternately_keep_their_gates_apart(streams):
    """two early-observation handles in one process, stepped in turns -- on one stream, and each on a stream of its own with the
    host never waiting in between: each handle's gate waits for its OWN sequence word and both stay bit-identical to lockstep
    reference handles"""
    import torch
    from img_env_amd.world import World
    n = 4096
    grid, params, layout = small_world(n, 40, seed=97, grid_size=400, res=0.25, clearance=0.6, n_obstacles=2)
    params = dict(params, output_guard="none")
    hs = [World(dict(params), grid) for _ in range(2)]
    ref = World(dict(params), grid)
    try:
        dev = ref.device
        rng = np.random.default_rng(31)
        acts = [torch.as_tensor(random_actions(rng, n), device=dev) for _ in range(30)]
        ref.reset(layout)
        want = []
        for s in range(30):
            ref.step(acts[s])
            want.append(ref.snapshot())
        sts = [torch.cuda.Stream(device=dev) for _ in range(2)] if streams == "two_streams" else [torch.cuda.current_stream(dev)] * 2
        for h, st in zip(hs, sts):
            with torch.cuda.stream(st):
                h.reset(layout)
        got = [[], []]
        for s in range(30):
            for q, (h, st) in enumerate(zip(hs, sts)):
                with torch.cuda.stream(st):
                    h.step(acts[s])
                    got[q].append({k: h.out[k].clone() for k in FIELDS})
        torch.cuda.synchronize()
        for q in range(2):
            for s in range(30):
                for k in FIELDS:
                    assert np.array_equal(got[q][s][k].cpu().numpy(), want[s][k], equal_nan=True), (streams, q, s, k)
    finally:
        ref.close()
        for h in hs:
            h.close()
