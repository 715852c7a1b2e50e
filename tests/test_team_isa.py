"""The persistent large-N kernel (csrc/spec_k_team.hip) keeps loads in flight across loop iterations with
inline-assembly loads and counted waits.  That is only sound if the compiler can never touch a load's
destination between the load and the wait that covers it -- hence LDS-DMA loads, which have no register
destination; tools/check_inflight.py compiles the file for gfx950 (hipcc cross-compiles without a GPU) and
scans the assembly of every instantiation."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelined_loads_are_lds_dma_and_nothing_touches_a_register_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "in-flight registers touched 0" in r.stdout
