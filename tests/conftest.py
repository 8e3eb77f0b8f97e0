import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _install_guard_allocator():
    """VN_GUARD_ALLOC=end|start (tests/test_gpu_guard.py sets it for its child processes): every torch allocation of this process becomes
    a guard block of csrc/devmem.hip — exact size, flanked by unmapped pages — and so does every allocation of the library itself.  Has to
    happen before the first CUDA allocation of the process."""
    import torch
    from vampnet_amd import _lib
    alloc = torch.cuda.memory.CUDAPluggableAllocator(_lib.LIB_PATH, "vn_guard_torch_alloc", "vn_guard_torch_free")
    torch.cuda.memory.change_current_allocator(alloc)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "reference: needs /root/reference (this container only)")
    if os.environ.get("VN_GUARD_ALLOC") and os.environ.get("VN_GUARD_TORCH", "1") != "0":
        _install_guard_allocator()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if os.environ.get("VN_GUARD_ALLOC"):
        import ctypes as C
        from vampnet_amd import _lib
        n, live, nbytes = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        mode = _lib.load().vn_guard_stats(C.byref(n), C.byref(live), C.byref(nbytes))
        terminalreporter.write_line(f"GUARD mode={mode} blocks={n.value} live={live.value} live_bytes={nbytes.value}")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)
