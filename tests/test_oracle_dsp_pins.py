"""Pins for the sub-blocks of the network restatements that an INSTALLED independent implementation can
check (VERDICT r2 next #6d).  The third-party graphs themselves stay unpinned (no pyannote.audio /
speechbrain / checkpoints here), but their signal-processing front ends are standard transforms:

* ECAPA's STFT power spectrum (oracle/ecapa_ref.py: torch.stft, 400-point Hamming window, hop 160,
  centred with zero padding) against scipy.signal.stft;
* its triangular mel filterbank against the textbook construction (triangles between mel-spaced
  edges evaluated with numpy.interp — a different formulation of the same definition);
* the DFT-as-GEMM matrices the HIP path multiplies with (diart_amd.weights) against numpy.fft;
* the LSTM restatement is pinned against torch.nn.LSTM elsewhere (test_oracle_golden.py)."""
import numpy as np
import torch


def test_stft_power_matches_scipy():
    from scipy import signal
    from oracle.ecapa_ref import HOP, N_FFT
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 16000 * 2 + 37)).astype(np.float32)
    window = torch.hamming_window(N_FFT)
    spec = torch.stft(torch.from_numpy(x), N_FFT, HOP, N_FFT, window, center=True, pad_mode="constant",
                      normalized=False, onesided=True, return_complex=True)
    power = (spec.real ** 2 + spec.imag ** 2).numpy()                       # (N, 201, T)
    win = signal.get_window("hamming", N_FFT, fftbins=True)
    assert np.allclose(win, window.numpy(), atol=1e-7)
    _, _, Z = signal.stft(x.astype(np.float64), fs=16000, window=win, nperseg=N_FFT, noverlap=N_FFT - HOP,
                          nfft=N_FFT, boundary="zeros", padded=False, return_onesided=True)
    Z = Z * win.sum()                                                       # undo scipy's spectrum scaling
    T = power.shape[2]
    assert Z.shape[2] >= T - 1
    n = min(T, Z.shape[2])
    ref = (np.abs(Z) ** 2)[:, :, :n]
    rel = np.abs(power[:, :, :n] - ref).max() / ref.max()
    assert rel < 1e-5, rel


def test_mel_filterbank_matches_the_textbook_triangles():
    from oracle.ecapa_ref import N_FFT, N_MELS, SAMPLE_RATE, mel_filterbank
    fb = mel_filterbank().numpy()                                           # (201, 80)
    to_mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
    to_hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    edges = to_hz(np.linspace(to_mel(0.0), to_mel(SAMPLE_RATE / 2), N_MELS + 2))
    freqs = np.linspace(0, SAMPLE_RATE // 2, N_FFT // 2 + 1)
    want = np.zeros_like(fb, dtype=np.float64)
    for m in range(N_MELS):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        # speechbrain's "triangular" filters are symmetric in Hz around the centre with the half-width of the
        # LOWER band (band = f[m+1] - f[m]); inside the textbook triangle both agree on the rising edge
        band = ce - lo
        want[:, m] = np.clip(1.0 - np.abs(freqs - ce) / band, 0.0, None)
        rising = (freqs >= lo) & (freqs <= ce)
        assert np.allclose(fb[rising, m], np.interp(freqs[rising], [lo, ce], [0.0, 1.0]), atol=2e-5)
    assert np.abs(fb - want).max() < 2e-5
    assert fb.shape == (201, 80) and (fb >= 0).all() and fb.max() <= 1.0 + 1e-6
    assert (fb.argmax(0)[1:] >= fb.argmax(0)[:-1]).all()                    # centres increase with the mel index


def test_dft_matrices_of_the_hip_path_match_numpy_fft():
    """diart_amd.weights.dft_matrices: the windowed DFT as one real GEMM operand (cos rows, then sin rows);
    a frame through it must be numpy's rfft of the windowed frame, and the power spectrum built from it
    must be the one torch.stft gives the oracle."""
    from scipy import signal
    from diart_amd.weights import dft_matrices
    m = dft_matrices().numpy()
    assert m.shape == (402, 400)
    rng = np.random.default_rng(1)
    frame = rng.standard_normal(400)
    win = signal.get_window("hamming", 400, fftbins=True)
    ref = np.fft.rfft(frame * win)
    out = m @ frame
    assert np.allclose(out[:201], ref.real, atol=1e-10) and np.allclose(out[201:], -ref.imag, atol=1e-10)
    assert np.allclose(out[:201] ** 2 + out[201:] ** 2, np.abs(ref) ** 2, rtol=1e-10, atol=1e-10)
