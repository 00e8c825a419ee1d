"""INTEGRATION.md section 1 on the reference's OWN base class: runs tests/golden/check_reference_mount.py where /root/reference exists
(the build container); skipped elsewhere - the reference never travels."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/mdl"), reason="needs the reference tree (build container only)")
def test_factory_classes_on_the_references_ntf():
    script = os.path.join(HERE, "golden", "check_reference_mount.py")
    if not os.path.exists(script):
        pytest.skip("check_reference_mount.py is not shipped to the GPU box")
    p = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "mount check passed" in out, out[-3000:]
    assert "FAIL" not in out
