"""The committed recipe regenerates the committed fixtures: tests/golden/make_golden.py is run (every generator) into
a scratch directory and each array of each .npz is compared with the file in tests/golden/ -- dtype, shape and value.
Needs the reference tree (/root/reference, this container only): skipped on the GPU box, where fixtures are data."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/ark"), reason="reference tree not present")
def test_make_golden_reproduces_every_committed_fixture(tmp_path):
    env = dict(os.environ, PXSOM_GOLDEN_OUT=str(tmp_path))
    proc = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, timeout=600)
    assert proc.returncode == 0, proc.stdout.decode(errors="replace")[-3000:]
    committed = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert len(committed) >= 21
    regenerated = {os.path.basename(p) for p in glob.glob(os.path.join(str(tmp_path), "*.npz"))}
    assert regenerated == {os.path.basename(p) for p in committed}
    for path in committed:
        want = np.load(path, allow_pickle=False)
        got = np.load(os.path.join(str(tmp_path), os.path.basename(path)), allow_pickle=False)
        assert set(want.files) == set(got.files), path
        for key in want.files:
            a, b = want[key], got[key]
            assert a.dtype == b.dtype and a.shape == b.shape, (path, key)
            assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), (os.path.basename(path), key)
