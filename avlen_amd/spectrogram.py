"""Device-side mirror of `SpectrogramSensor.compute_spectrogram` (soundspaces/tasks/nav.py:88-101):
log1p( 4x4 block mean of |STFT(n_fft=512, hop=160, win=400)| ) per binaural channel, batched over environments.

The reference computes this on the CPU per environment with librosa + scikit-image (both un-pinned and absent here: the published
definitions are restated, parity unpinned -- oracle/restate_audio.py is the checker).  `pool=4` yields the reference's (65, 26, 2)
map, `pool=1` the full-resolution (257, 101, 2) map that BASELINE.json's synthetic observations use.
"""
import math
import torch

from . import _lib as L
from .engine import P

N_FFT, HOP, WIN = 512, 160, 400


class Spectrogram:
    def __init__(self, device="cuda", n_fft=N_FFT, hop_length=HOP, win_length=WIN, pad_mode="reflect"):
        assert pad_mode in ("reflect", "constant"), "librosa.stft pad_mode: 'reflect' (< 0.10 default) or 'constant' (>= 0.10)"
        self.n_fft, self.hop, self.reflect, self.device = n_fft, hop_length, int(pad_mode == "reflect"), torch.device(device)
        n = torch.arange(win_length, dtype=torch.float64)
        hann = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / win_length)               # periodic ('fftbins=True') Hann
        lp = (n_fft - win_length) // 2
        w = torch.zeros(n_fft, dtype=torch.float64)
        w[lp:lp + win_length] = hann                                               # librosa pads the window to n_fft, centred
        k = torch.arange(n_fft // 2 + 1, dtype=torch.float64).view(-1, 1)
        ang = 2.0 * math.pi * k * torch.arange(n_fft, dtype=torch.float64).view(1, -1) / n_fft
        basis = torch.cat([torch.cos(ang), -torch.sin(ang)], 0)
        self.window = w.float().to(self.device).contiguous()
        self.basis = basis.float().to(self.device).contiguous()
        self._ws = None

    def __call__(self, audio, pool=4):
        """audio (B, 2, L) fp32 on the device -> (B, H, W, 2) with (H, W) = (65, 26) for pool=4 / (257, 101) for pool=1 at L=16000."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 3 and audio.shape[1] == 2
        a = audio.contiguous()
        B, _, Ln = a.shape
        nb = L.lib.avlen_spectrogram_workspace_bytes(B, Ln, self.n_fft, self.hop)
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty(nb, dtype=torch.uint8, device=a.device)
        F, NB = 1 + Ln // self.hop, self.n_fft // 2 + 1
        out = torch.empty(B, (NB + pool - 1) // pool, (F + pool - 1) // pool, 2, device=a.device)
        L.call("avlen_spectrogram", P(a), B, Ln, P(self.window), P(self.basis), self.n_fft, self.hop, pool, self.reflect, P(out),
               P(self._ws), nb, L.stream())
        return out

    def compute_spectrogram(self, audio_data):
        """Reference signature: (2, L) array-like of one environment -> (65, 26, 2) numpy array."""
        a = torch.as_tensor(audio_data, dtype=torch.float32).to(self.device).unsqueeze(0)
        return self(a, 4)[0].cpu().numpy()
