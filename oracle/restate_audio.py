"""CPU restatement of SpectrogramSensor.compute_spectrogram (soundspaces/tasks/nav.py:88-101) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in librosa.stft and skimage.measure.block_reduce, both un-pinned in the reference's setup.py
(:35, :44) and absent from this image; their published definitions are restated with numpy's FFT:
  librosa.stft(y, n_fft=512, hop_length=160, win_length=400): window = periodic Hann(400) zero-padded to 512 (centred);
    center=True pads the signal by n_fft//2 on both sides (pad_mode 'reflect' before librosa 0.10, 'constant' from 0.10 on);
    frame f starts at sample f*hop of the padded signal; n_frames = 1 + len(y)//hop; output (1 + n_fft/2, n_frames) complex.
  block_reduce(x, (4, 4), np.mean): pads x with zeros up to a multiple of the block size, then means every block."""
import numpy as np


def stft_mag(signal, n_fft=512, hop=160, win=400, pad_mode="reflect"):
    n = np.arange(win)
    w = np.zeros(n_fft)
    lp = (n_fft - win) // 2
    w[lp:lp + win] = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win)
    y = np.pad(np.asarray(signal, dtype=np.float64), n_fft // 2, mode=pad_mode)
    nf = 1 + len(signal) // hop
    frames = np.stack([y[f * hop:f * hop + n_fft] * w for f in range(nf)], 1)          # (n_fft, frames)
    return np.abs(np.fft.rfft(frames, axis=0))                                           # (257, frames)


def block_mean(x, b=4):
    H, W = x.shape
    Hp, Wp = -(-H // b) * b, -(-W // b) * b
    p = np.zeros((Hp, Wp), dtype=x.dtype)
    p[:H, :W] = x
    return p.reshape(Hp // b, b, Wp // b, b).mean((1, 3))


def compute_spectrogram(audio, pool=4, pad_mode="reflect"):
    """audio (2, L) -> (65, 26, 2) for pool=4 (the reference), (257, 101, 2) for pool=1."""
    ch = [np.log1p(block_mean(stft_mag(audio[c], pad_mode=pad_mode), pool) if pool > 1 else stft_mag(audio[c], pad_mode=pad_mode))
          for c in range(2)]
    return np.stack(ch, -1).astype(np.float32)
