"""Stamped blocks (slam-tricks_amd/csrc/common.hpp): how a kernel hands a few doubles to the polling host through mapped host
memory.  Round 6 measured that a sequence number written BEHIND the payload can be seen by the host ahead of payload in another
cache line (tools/dbg/tri_repeat.py: 23 of 300 runs of the reference's 600 per-landmark solves ended somewhere else), so every line
carries its stamp and a check word and the host validates what it reads.  Here: the host-side validator against hand-packed lines,
torn ones included (no device needed)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_validator_refuses_torn_lines(tmp_path):
    exe = str(tmp_path / "test_stamped_block")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "test_stamped_block.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "stamped_block ok" in p.stdout, p.stdout[-2000:]
