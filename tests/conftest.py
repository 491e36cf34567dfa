import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SHARD_JOB = {}  # the two-process HIP shard run (tests/shard_ranks.py): started here, judged by tests/test_gpu_shard_processes.py
RCCL_JOB = {}   # N ranks on N GPUs with the library's own ncclAllGather (boxes with >= 2 devices): tests/test_gpu_rccl_ranks.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_tests_selected(config):
    expr = (config.getoption("markexpr", "") or "").strip()
    if "not gpu" in expr:
        return False
    try:
        import torch
        return torch.cuda.device_count() > 0  # (counting devices does not initialise the GPU)
    except Exception:
        return False


def pytest_collection_finish(session):
    """The ranks of the two-process shard test are child processes, and nothing may be started from a process that already holds
    a GPU context on this pool: so their launcher (which never touches the GPU itself) is started now -- the tests are collected,
    none has run, no module touches the GPU at import -- and runs underneath the other tests."""
    if os.environ.get("IMGENV_NO_SHARD_JOB") or not _gpu_tests_selected(session.config):
        return
    if session.config.getoption("collectonly", False):
        return
    if any("test_gpu_shard_processes" in it.nodeid for it in session.items):
        out = tempfile.mkdtemp(prefix="imgenv_shard_")
        SHARD_JOB["dir"] = out
        SHARD_JOB["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_ranks.py"), "launch", out],
                                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import torch
    n_dev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if n_dev >= 2 and any("test_gpu_rccl_ranks" in it.nodeid for it in session.items):
        out = tempfile.mkdtemp(prefix="imgenv_rccl_")
        RCCL_JOB["dir"], RCCL_JOB["ranks"] = out, min(n_dev, 8)
        RCCL_JOB["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_ranks.py"), "launch_rccl", out, str(min(n_dev, 8))],
                                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def pytest_sessionfinish(session, exitstatus):
    for job in (SHARD_JOB, RCCL_JOB):
        pr = job.get("proc")
        if pr is not None and pr.poll() is None:
            try:
                pr.wait(timeout=600)
            except subprocess.TimeoutExpired:
                pr.kill()


@pytest.fixture(scope="session")
def oracle_lib():
    """builds oracle/liboracle.so (and oracle/_ref when /root/reference exists)"""
    import oracle_binding
    oracle_binding.build_oracle()
    return oracle_binding.load_oracle()
