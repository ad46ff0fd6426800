"""GPU: the hashes of a clip do not depend on the batch it is fingerprinted in.

Large batches switch peak_pick to long time segments (PK_SEG_LONG: fewer halo frames re-read) and spread the
frames of a clip over several workgroups; small batches use short segments.  Both must give the same
(key32, t1) sequence, which the small-batch parity tests pin against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_large_batch_equals_single_clip_calls():
    import shazam_amd as S
    from oracle import synth
    ctx = S.get_context(0)
    n_clips, n = 300, 45 * 44100          # 969 frames per clip: two long segments; 290,700 frames in the batch
    pcm = ctx.synth_pcm(99, 0, n_clips, n, 2500, 3000)
    off = np.arange(n_clips + 1, dtype=np.uint64) * n
    k, t1, ho, _ = ctx.fingerprint_batch(pcm, off, pcm_device=True)
    pcm.free()
    assert len(ho) == n_clips + 1 and ho[-1] == len(k)
    for c in (0, 1, 137, 299):
        x = synth.synth_clip(99, c, n, 2500, 3000)                       # numpy twin of the device generator
        k1, t11, ho1, _ = ctx.fingerprint_batch(x, np.array([0, n], np.uint64))
        a, b = int(ho[c]), int(ho[c + 1])
        assert b - a == len(k1) > 1000
        assert np.array_equal(k[a:b], k1) and np.array_equal(t1[a:b], t11)
