"""CPU tier: the wave-specialised implicit-GEMM kernel on the host SIMT simulator (child process, see
tests/ws_emu_cases.py for why)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ws_kernel_on_simulator():
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    env = dict(os.environ, DPC_IGEMM_WS_MINROWS="1", DPC_IGEMM_WS_GM="2", DPC_HALO_WS_GM="3", DPC_IGEMM_GM_CAP="32", DPC_IGEMM_WS_PAR_MINCO="64", DPC_WSD_MINPLANES="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ws_emu_cases.py")], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0 and "ws cases ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
