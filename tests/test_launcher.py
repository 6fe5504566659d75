"""CPU tier: the pre-GPU launcher of conftest.py (the GPU tier forks its rank processes through it, never from the
GPU-initialised pytest process: DESIGN.md 9.7)."""
import os
import sys

import conftest


def test_clean_launcher_runs_commands_and_reports_failures():
    la = conftest.CleanLauncher()
    try:
        rc, out, err = la.run([sys.executable, "-c", "import os, sys; print(os.environ['DPC_T']); sys.stderr.write('e'); sys.exit(3)"],
                              dict(os.environ, DPC_T="through the launcher"), 30)
        assert (rc, out.strip(), err) == (3, "through the launcher", "e")
        rc, out, err = la.run([sys.executable, "-c", "print(1)"], dict(os.environ), 30)   # the helper serves more than one request
        assert rc == 0 and out.strip() == "1"
        rc, _, err = la.run(["/nonexistent/binary"], dict(os.environ), 5)
        assert rc == -999 and "No such file" in err
        rc, _, err = la.run([sys.executable, "-c", "import time; time.sleep(5)"], dict(os.environ), 0.2)
        assert rc == -999 and "TimeoutExpired" in err
    finally:
        la.close()
    assert la.p.poll() is not None


def test_launcher_is_only_started_for_the_gpu_tier(clean_launcher, request):
    expr = request.config.getoption("-m", default="") or ""
    assert (clean_launcher is not None) == ("gpu" in expr and "not gpu" not in expr)
