"""CPU tier: the package's runtime-flag default (dpc_amd/__init__.py) is a default, not an override."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(env_extra):
    env = {k: v for k, v in os.environ.items() if k != "HIP_FORCE_DEV_KERNARG"}
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", "import os, dpc_amd; print(os.environ.get('HIP_FORCE_DEV_KERNARG'))"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip()


def test_kernarg_flag_is_set_before_hip_initialises_unless_the_caller_chose():
    assert _probe({}) == "1"
    assert _probe({"HIP_FORCE_DEV_KERNARG": "0"}) == "0"
