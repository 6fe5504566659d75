import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# ---- a launcher process started BEFORE this process touches the GPU.
# The RCCL tests start their ranks as child processes (torch.distributed.run).  Forking from the pytest process once the HIP / HSA
# runtime is up (hundreds of GB of device mappings, runtime threads) segfaulted inside subprocess.Popen about once in five runs of the GPU
# tier on the MI355X boxes (faulthandler: subprocess.py `__init__` <- tests/test_graph_rccl_gpu.py `_run_ranks`; round 4).  So the children
# are forked by a small helper that is itself forked at configure time, while this process is still an ordinary Python process.
_LAUNCHER_CODE = r"""
import json, subprocess, sys
for line in sys.stdin:
    req = json.loads(line)
    try:
        r = subprocess.run(req["argv"], env=req["env"], capture_output=True, text=True, timeout=req["timeout"])
        out = {"returncode": r.returncode, "stdout": r.stdout[-20000:], "stderr": r.stderr[-20000:]}
    except Exception as e:
        out = {"returncode": -999, "stdout": "", "stderr": repr(e)}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
"""
_launcher = None


class CleanLauncher:
    def __init__(self):
        import subprocess
        self.p = subprocess.Popen([sys.executable, "-u", "-c", _LAUNCHER_CODE], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    def run(self, argv, env, timeout):
        """subprocess.run(argv, env=env, capture_output=True, text=True, timeout=timeout) in the helper: (returncode, stdout, stderr)"""
        import json
        self.p.stdin.write(json.dumps({"argv": list(argv), "env": dict(env), "timeout": timeout}) + "\n")
        self.p.stdin.flush()
        line = self.p.stdout.readline()
        if not line:
            raise RuntimeError("the launcher process died")
        out = json.loads(line)
        return out["returncode"], out["stdout"], out["stderr"]

    def close(self):
        try:
            self.p.stdin.close()
            self.p.wait(timeout=10)
        except Exception:
            self.p.kill()


def _cpu_tier_workers(config, expr):
    """The CPU tier (`-m "not gpu"`) spends its time in the host-side SIMT simulator, one core per test: spread it over
    pytest-xdist workers unless the caller chose a worker count (-n ...) or DPC_TEST_WORKERS=0 asks for one process.  The GPU tier
    always runs in ONE process (one device, timing-sensitive tests, the pre-GPU launcher below)."""
    if hasattr(config, "workerinput") or "not gpu" not in expr:
        return 0
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) is not None:
        return 0
    if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
        return 0
    want = os.environ.get("DPC_TEST_WORKERS")
    n = int(want) if want is not None else min(6, max(1, (os.cpu_count() or 2) - 2))
    return n if n > 1 else 0


def _build_once():
    """`make all emu` under an exclusive file lock: the test modules call `make emu` themselves, several xdist workers at a time --
    after this that is a no-op everywhere, and only one process ever writes the objects"""
    import fcntl
    import subprocess
    with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-s", "-j8", "all", "emu"], cwd=ROOT, check=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def pytest_configure(config):
    global _launcher
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    expr = config.getoption("-m", default="") or ""
    if "not gpu" in expr and not getattr(config.option, "collectonly", False):
        _build_once()   # controller first, then every worker finds nothing to do
    n = _cpu_tier_workers(config, expr)
    if n:   # what `-n N` does in xdist's pytest_cmdline_main; its own pytest_configure (trylast) then starts the workers
        config.option.numprocesses = n
        config.option.dist = "load"
        config.option.tx = ["popen"] * n
    if "gpu" in expr and "not gpu" not in expr and _launcher is None:   # the GPU tier: nothing has initialised HIP yet
        _launcher = CleanLauncher()


@pytest.fixture(scope="session")
def clean_launcher():
    """the pre-GPU launcher process (None outside the GPU tier: callers fall back to subprocess.run)"""
    return _launcher


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_unconfigure(config):
    global _launcher
    if _launcher is not None:
        _launcher.close()
        _launcher = None


def pytest_sessionfinish(session, exitstatus):
    """Engines (their side streams, captured graphs, scratch pools) go while the HIP runtime is still up and idle: collecting
    them during interpreter shutdown, with work possibly still queued, is where an intermittent abort at exit was seen once."""
    import gc
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
