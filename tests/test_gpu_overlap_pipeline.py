"""GPU: the two-stream extraction pipeline (SHZ_OVERLAP_SPLIT >= 2: STFT of sub-batch i+1 on a second stream beside
peak picking of sub-batch i) gives the hashes of the sequential pass.  The switch is read once per process, so the
pipelined run happens in a child process; both fingerprint the same device-generated clips twice in a row (the second
call starts while nothing has synchronised the generator of its input: the second stream has to wait for it)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, hashlib
import numpy as np
sys.path.insert(0, %r)
from shazam_amd import _ffi
ctx = _ffi.Context(0)
n, nc = 30 * 44100, 130
off = np.arange(nc + 1, dtype=np.uint64) * n
pcm = ctx.alloc(nc * n * 2)
h = hashlib.sha256()
for rnd in range(3):
    ctx.synth_pcm(99, 1000 * rnd, nc, n, 3000, 2500, out=pcm)       # not synchronised: the pass must order itself behind it
    k, t1, ho, cnt = ctx.fingerprint_batch(pcm, off, pcm_device=True)
    for a in (k, t1, ho):
        h.update(np.ascontiguousarray(a).tobytes())
print(h.hexdigest(), ctx.extract_stats()["f64_passes"])
"""


def _run(split, dual=0):
    env = dict(os.environ, SHZ_OVERLAP_SPLIT=str(split), SHZ_DUAL=str(dual))
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip().split()


def test_pipelined_pass_equals_sequential_pass():
    seq = _run(0)
    for split in (2, 5):
        assert _run(split) == seq, split
    assert seq[1] == "0"


def test_dual_pass_equals_single_pass():
    """SHZ_DUAL=1: the two halves of the batch as two passes on two contexts, entries of the second appended behind the
    first's -- same hashes, same offsets (the child fingerprints 130 x 30 s = 83,720 frames... below the dual threshold
    of 131,072 frames, so a second child with 260 clips crosses it)."""
    child_big = CHILD.replace("n, nc = 30 * 44100, 130", "n, nc = 30 * 44100, 260")
    outs = []
    for dual in (0, 1):
        env = dict(os.environ, SHZ_OVERLAP_SPLIT="0", SHZ_DUAL=str(dual))
        out = subprocess.run([sys.executable, "-c", child_big % ROOT], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out.stdout.strip().split())
    assert outs[0] == outs[1]
