"""NOT collected by a normal run (the name does not match test_*.py): the two cases tests/test_suite_order_cpu.py feeds to the `isolated` marker —
one body that kills its interpreter the way the HIP runtime does (abort() from a thread Python does not know), one that passes."""
import ctypes
import os
import threading

import pytest


@pytest.mark.isolated
def test_body_that_aborts_from_a_native_thread():
    assert os.environ.get("ZKPOR_ISOLATED_CHILD") == "1"      # only ever runs in the child
    t = threading.Thread(target=lambda: ctypes.CDLL(None).abort())
    t.start()
    t.join()


@pytest.mark.isolated
def test_body_that_passes():
    assert os.environ.get("ZKPOR_ISOLATED_CHILD") == "1"
