"""-m gpu: the memory-safety harness (round 6; VERDICT r5 "make memory safety deterministic").

hipMalloc sub-allocates from large mapped blocks, so a kernel that reads a few KiB past its buffer is silent almost always and, once in a
long while, an abort nobody can reproduce (profiles/r05_pytest_gpu_one_aborted_run.txt).  Here the same tests run in CHILD processes
in which every device allocation — torch's, through the pluggable allocator, and the library's own (csrc/devmem.hip) — is an exact-size
guard block flanked by unmapped pages, once with the buffer's END against the unmapped page and once with its START on a page start.
Any access outside a buffer is a GPU memory fault that kills the child; a green child therefore proves every access of every launch of
those tests in range.  The first test proves the harness itself: a deliberate one-word over-read must kill its child."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

POKE = r"""
import ctypes as C, sys, torch
from vampnet_amd import _lib
lib = _lib.load()
p = C.c_void_p()
assert lib.vn_guard_alloc(1000, int(sys.argv[1]), C.byref(p)) == 0
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
rc = lib.vn_guard_poke(p, int(sys.argv[2]), int(sys.argv[3]), sink.data_ptr(), None)
torch.cuda.synchronize()
print("POKE_RETURNED", rc, flush=True)
"""


def _child(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_harness_kills_a_process_that_reads_one_word_past_its_buffer():
    inside = _child(["-c", POKE, "1", "996", "0"])                  # last word of a 1000-byte block (1008 after the 16-byte rounding)
    assert inside.returncode == 0 and "POKE_RETURNED 0" in inside.stdout, inside.stderr[-2000:]
    for mode, off, write in ((1, 1008, 0), (1, 1008, 1), (2, -4, 0)):   # first word behind an end block / in front of a start block
        out = _child(["-c", POKE, str(mode), str(off), str(write)])
        assert out.returncode != 0 and "POKE_RETURNED 0" not in out.stdout, (mode, off, out.stdout[-500:], out.stderr[-500:])
    again = _child(["-c", POKE, "2", "0", "1"])                     # the device is fine afterwards
    assert again.returncode == 0, again.stderr[-2000:]


# what runs under guard pages inside the driver's `-m gpu` run, in both alignments: the ragged cases (models of the three precisions,
# generate, every attention decomposition, the 50-call training attention, ragged training steps) and EVERY single-op kernel test — 423
# tests, ~80-90 s per mode.  The WHOLE suite under guard pages (710 tests per mode, 29 145 guard blocks, 11 minutes each) is
# scripts/gpu_guard_full.sh; its logs are profiles/r06_guard_full_{end,start}.log.
# (+ round 6's operand modes of the GEMM: the channels-on-rows convolutions with their per-tap gather; the token-major dW GEMMs are in
# test_gpu_kernels.py, the training step with its side stream in the guard cases)
GUARDED = ["tests/test_gpu_guard_cases.py", "tests/test_gpu_kernels.py",
           "tests/test_gpu_codec.py::test_conv_channels_on_rows_equals_the_column_tile_form"]


@pytest.mark.parametrize("mode", ["end", "start"])
def test_guarded_child_suite(mode):
    if os.environ.get("VN_GUARD_ALLOC"):
        pytest.skip("this IS a guarded child")
    out = _child(["-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider"] + GUARDED, {"VN_GUARD_ALLOC": mode}, timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    m = re.search(r"GUARD mode=(\d) blocks=(\d+)", out.stdout)
    assert m and int(m.group(1)) == (1 if mode == "end" else 2) and int(m.group(2)) > 100, tail     # the children really ran on guard blocks
