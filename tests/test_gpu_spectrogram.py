"""Device spectrogram (soundspaces/tasks/nav.py:88-101) vs the numpy/FFT restatement of librosa.stft + skimage
block_reduce (oracle/restate_audio.py; parity unpinned against the reference: both packages are un-pinned and absent)."""
import numpy as np
import pytest
import torch

import restate_audio as ra
from avlen_amd.spectrogram import Spectrogram

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pad_mode", ["reflect", "constant"])
def test_spectrogram_matches_fft_restatement(pad_mode):
    rs = np.random.RandomState(3)
    B, Ln = 3, 16000
    t = np.arange(Ln) / 16000.0
    audio = 0.3 * rs.randn(B, 2, Ln) + np.sin(2 * np.pi * 440.0 * t)[None, None] * np.array([1.0, 0.5])[None, :, None]
    audio[1, :, 8000:] = 0.0                                    # a silent tail (episode end)
    audio[2] = 0.0                                              # silence: log1p(0) = 0 everywhere
    sp = Spectrogram(pad_mode=pad_mode)
    a = torch.from_numpy(audio.astype(np.float32)).cuda()
    lo, hi = sp(a, 4).cpu().numpy(), sp(a, 1).cpu().numpy()
    assert lo.shape == (B, 65, 26, 2) and hi.shape == (B, 257, 101, 2)      # the reference's sensor shape / BASELINE's shape
    for b in range(B):
        ref4 = ra.compute_spectrogram(audio[b].astype(np.float32), 4, pad_mode)
        ref1 = ra.compute_spectrogram(audio[b].astype(np.float32), 1, pad_mode)
        np.testing.assert_allclose(lo[b], ref4, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(hi[b], ref1, rtol=1e-4, atol=2e-4)
    assert float(np.abs(lo[2]).max()) == 0.0
    # reference signature: one environment's (2, L) array in, (65, 26, 2) numpy out
    one = sp.compute_spectrogram(audio[0])
    np.testing.assert_allclose(one, lo[0], rtol=0, atol=0)
    # the unit-amplitude call the sensor uses to size its observation space (nav.py:79)
    assert sp.compute_spectrogram(np.ones((2, 16000))).shape == (65, 26, 2)
