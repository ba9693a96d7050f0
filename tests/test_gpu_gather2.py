"""The C ABI's PCM gather with TWO ranks: the N > 1 data path and its failure behaviour on the one GPU a box has, through the
shared-directory test transport tools/stub_rccl_shm.c in place of RCCL. See tools/gather2_rank.py for what each rank checks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_gather_every_row_and_fail_together(tmp_path):
    stub = str(tmp_path / "libstub_rccl_shm.so")
    cc = subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tools", "stub_rccl_shm.c"), "-o", stub,
                         "-L/opt/rocm/lib", "-lamdhip64"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    d = tmp_path / "xchg"
    d.mkdir()
    env = dict(os.environ, VITS_RCCL_LIB=stub, STUB_RCCL_DIR=str(d), STUB_RCCL_LOG=str(tmp_path / "stub.log"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gather2_rank.py"), str(r), "2", str(tmp_path / "uid")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung")
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "gather2 ok rank %d" % r in so, (r, so[-2000:], se[-4000:])
    log = open(tmp_path / "stub.log").read().split("\n")
    # the failing exchanges stopped after the FIRST collective on both ranks: per element type 2 good exchanges x 2 collectives + 4 failing x 1, no abort
    assert sum(ln.startswith("allgather 0") for ln in log) == sum(ln.startswith("allgather 1") for ln in log) == 2 * (2 * 2 + 3 + 1)
    assert not any(ln.startswith("abort") for ln in log)
