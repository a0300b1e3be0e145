"""Runs tests/test_hip_parity._rccl_worker in THIS process under faulthandler (a crash there is a bare SIGSEGV under mp.spawn)."""
import faulthandler
import os
import sys

faulthandler.enable()
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(root, "tests"), root, os.path.join(root, "handheld-multi-frame-super-resolution_amd")]
import test_hip_parity as t  # noqa: E402

t._rccl_worker(0, 29517, "/tmp/o.npz")
print("worker ok")
