// Binaural spectrogram on the device: SpectrogramSensor.compute_spectrogram (soundspaces/tasks/nav.py:88-101)
//   |STFT(n_fft 512, hop 160, hann window 400 zero-padded to 512, centred frames)| -> 4x4 block mean (zero-padded edges) -> log1p
// librosa.stft and skimage.measure.block_reduce are un-pinned third-party packages absent from this image (setup.py:35,44): the
// published definitions are restated (oracle/restate_audio.py is the numpy/FFT side) and parity is UNPINNED against the reference.
//
// The transform is a real DFT of 400 non-zero samples per frame: frames (B*2*F rows, windowed, K = 512) times a [cos | -sin] basis
// (2*257 columns) on the fp32 MFMA GEMM of the training path (6.8 GFLOP for 64 envs -- not worth an FFT), then one fused pass
// for magnitude + pooling + log1p.  pool = 4 gives the reference's (65, 26) map; pool = 1 the full (257, 101) map BASELINE.json's
// synthetic observations use, so both resolutions come from the same waveform.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include <math.h>

namespace {

// frames[(b*2 + ch)*F + f][n] = w[n] * x_padded[f*hop + n],  x_padded = signal centred-padded by nfft/2 (reflect or zeros)
__global__ void stft_frames_kernel(const float* __restrict__ audio, const float* __restrict__ window, float* __restrict__ frames,
                                   long rows, int L, int F, int nfft, int hop, int reflect) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nfft) return;
  const int n = (int)(i % nfft); const long r = i / nfft;
  const int f = (int)(r % F); const long sig = r / F;
  int t = f * hop + n - nfft / 2;
  float v = 0.f;
  if (t < 0) { if (reflect) t = -t; else t = -1; }
  else if (t >= L) { if (reflect) t = 2 * (L - 1) - t; else t = -1; }
  if (t >= 0 && t < L) v = audio[sig * L + t];
  frames[i] = v * window[n];
}

// spec[(b*2+ch)*F + f][0..NB) = re, [NB..2NB) = im  ->  out[b][i][j][ch] = log1p(mean over the pool x pool block of |X[k][f]|),
// blocks past the edge padded with zeros (skimage block_reduce, cval = 0)
__global__ void stft_pool_kernel(const float* __restrict__ spec, float* __restrict__ out, int B, int F, int NB, int pool, int OH,
                                 int OW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * OH * OW * 2) return;
  const int ch = (int)(i & 1); long r = i >> 1;
  const int ow = (int)(r % OW); r /= OW;
  const int oh = (int)(r % OH); const long b = r / OH;
  float s = 0.f;
  for (int dk = 0; dk < pool; dk++) {
    const int k = oh * pool + dk;
    if (k >= NB) break;
    for (int df = 0; df < pool; df++) {
      const int f = ow * pool + df;
      if (f >= F) break;
      const float* p = spec + ((b * 2 + ch) * F + f) * (long)(2 * NB);
      const float re = p[k], im = p[NB + k];
      s += sqrtf(re * re + im * im);
    }
  }
  out[i] = log1pf(s / (float)(pool * pool));
}

}  // namespace

extern "C" size_t avlen_spectrogram_workspace_bytes(int B, int L, int nfft, int hop) {
  const long F = 1 + L / hop, rows = (long)B * 2 * F, NB = nfft / 2 + 1;
  return (size_t)rows * nfft * 4 + (size_t)rows * 2 * NB * 4 + (96u << 20) + 4096;
}

extern "C" int avlen_spectrogram(const float* audio, int B, int L, const float* window, const float* basis, int nfft, int hop,
                                 int pool, int reflect, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!audio || !window || !basis || !out || B <= 0 || L <= 0 || nfft <= 0 || (nfft & 7) || hop <= 0 || pool <= 0) return AVLEN_ERR_ARG;
  if (ws_bytes < avlen_spectrogram_workspace_bytes(B, L, nfft, hop)) return AVLEN_ERR_WS;
  const int F = 1 + L / hop, NB = nfft / 2 + 1;
  const long rows = (long)B * 2 * F;
  WsBump w(ws, ws_bytes);
  float* frames = w.take<float>((size_t)rows * nfft);
  float* spec = w.take<float>((size_t)rows * 2 * NB);
  void* gws = w.take<char>(96u << 20);
  const long tot = rows * nfft;
  hipLaunchKernelGGL(stft_frames_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, audio, window, frames, rows, L, F, nfft,
                     hop, reflect);
  // spec = frames * basis^T  (basis rows: cos k = 0..NB-1, then -sin k = 0..NB-1; fp32 operands, exact fp32 fma chains)
  int rc = avlen_gemm(frames, nfft, 0, basis, nfft, 0, spec, 2 * NB, nullptr, nullptr, 0, (int)rows, 2 * NB, nfft, 0,
                      AVLEN_PREC_FP32, 1, 0.f, gws, 96u << 20, st);
  if (rc != AVLEN_OK) return rc;
  const int OH = (NB + pool - 1) / pool, OW = (F + pool - 1) / pool;
  const long no = (long)B * OH * OW * 2;
  hipLaunchKernelGGL(stft_pool_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, spec, out, B, F, NB, pool, OH, OW);
  return avlen_launch_status();
}
