"""Host runtime (checkpoint bundle, idx reader, parameter service) under AddressSanitizer + UBSan and under
ThreadSanitizer.  The self-test binary is built from the same sources as libdtf_runtime.so; the Hogwild update of
the parameter service is the one intentional race and is excluded from TSan by attribute in the source."""
import os
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "distributed_tensorflow_amd", "csrc")


@pytest.mark.slow
@pytest.mark.parametrize("target", ["selftest-asan", "selftest-tsan"])
def test_runtime_selftest_is_clean_under_sanitizers(target):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host toolchain")
    r = subprocess.run(["make", "-C", CSRC, target], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "runtime selftest: ok" in out
    assert "Sanitizer" not in out          # no ASan / UBSan / TSan report of any kind
