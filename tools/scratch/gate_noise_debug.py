import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from img_env_amd.world import World
from scenarios import random_actions, small_world
n = 8192
grid, params, layout = small_world(n, 60, seed=95, grid_size=400, res=0.25, clearance=0.6, n_obstacles=2)
mode = sys.argv[1] if len(sys.argv) > 1 else "stream"
a, b = World(dict(params), grid), World(dict(params), grid)
a.reset(layout); b.reset(layout)
rng = np.random.default_rng(29)
dev = a.device
acts = [torch.as_tensor(random_actions(rng, n), device=dev) for _ in range(10)]
want = [b.snapshot()]
for s in range(10):
    b.step(acts[s]); want.append(b.snapshot())
noise = torch.cuda.Stream(device=dev)
x = torch.ones(256 * 1024 * 1024 // 4, device=dev)
K = "ped_vector_states"
for s in range(10):
    if mode != "quiet":
        with torch.cuda.stream(noise):
            for _ in range(40):
                x.mul_(1.0000001)
    t0 = time.perf_counter()
    a.step(acts[s])
    if mode == "device":
        torch.cuda.synchronize()
    else:
        torch.cuda.current_stream(dev).synchronize()
    dt = time.perf_counter() - t0
    g1 = a.out[K].cpu().numpy()
    torch.cuda.synchronize()
    g2 = a.out[K].cpu().numpy()
    print(mode, "step", s, "%.2f ms" % (1e3 * dt), "first read == want:", np.array_equal(g1, want[s + 1][K]), "== previous:", np.array_equal(g1, want[s][K]),
          "| after device sync == want:", np.array_equal(g2, want[s + 1][K]), "rows differing first read:", int((g1 != want[s + 1][K]).any(axis=1).sum()))
a.close(); b.close()
