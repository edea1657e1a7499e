"""Pins for the sub-blocks of the network restatements that an INSTALLED independent implementation can
check (VERDICT r2 next #6d).  The third-party graphs themselves stay unpinned (no pyannote.audio /
speechbrain / checkpoints here), but their signal-processing front ends are standard transforms:

* ECAPA's STFT power spectrum (oracle/ecapa_ref.py: torch.stft, 400-point Hamming window, hop 160,
  centred with zero padding) against scipy.signal.stft;
* its triangular mel filterbank against the textbook construction (triangles between mel-spaced
  edges evaluated with numpy.interp — a different formulation of the same definition);
* the DFT-as-GEMM matrices the HIP path multiplies with (diart_amd.weights) against numpy.fft;
* SincNet's parametric filter bank (oracle AND diart_amd.weights.sinc_filters) against its definition — the windowed
  quadrature band-pass of each (low, high) pair — evaluated by numerical quadrature, and its frequency response;
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


def test_sinc_filter_bank_is_the_windowed_quadrature_band_pass_of_its_parameters():
    """SincNet's first layer (asteroid ``ParamSincFB`` behind pyannote's ``SincNet``, third party; reached from
    /root/reference/src/diart/models.py:133, :262) is defined by 40 (low, high) pairs: filter i is the band-pass whose
    frequency response is 1 on [low_i, high_i] — an even ("cos") and an odd ("sin") impulse response, i.e.
    ``h_c[n] = int 2 cos(2 pi f n) df`` and ``h_s[n] = int 2 sin(2 pi f n) df`` over the band (Ravanelli & Bengio 2018,
    eq. 4 - 6) — truncated to 251 taps under a Hamming window and scaled by 1 / (2 band).  Here that definition is
    evaluated by numerical quadrature in float64, with no closed form, and both restatements of the library's closed
    form — the oracle's (``ParamSincFBRef.filters``) and the product's (``weights.sinc_filters``, what the HIP bank is
    packed from) — must reproduce it, at the mel-spaced initial parameters and at random ones; the DFT of a filter must
    be a band-pass where its parameters say."""
    import torch
    from diart_amd.weights import sinc_filters
    from oracle.models_ref import ParamSincFBRef

    sr, ks, half = 16000.0, 251, 125
    n = np.arange(-half, half + 1, dtype=np.float64)
    win = np.hamming(ks)
    win[half] = 1.0                                               # the centre tap is the band itself, unwindowed
    rng = np.random.default_rng(3)
    for trial in range(3):
        fb = ParamSincFBRef()
        if trial:
            with torch.no_grad():
                fb.low_hz_.copy_(torch.from_numpy(rng.uniform(-200.0, 6000.0, (40, 1)).astype("float32")))
                fb.band_hz_.copy_(torch.from_numpy(rng.uniform(-100.0, 1500.0, (40, 1)).astype("float32")))
        low = 50.0 + np.abs(fb.low_hz_.detach().numpy().astype(np.float64)[:, 0])
        high = np.clip(low + 50.0 + np.abs(fb.band_hz_.detach().numpy().astype(np.float64)[:, 0]), 50.0, sr / 2)
        want = np.zeros((80, ks))
        q = (np.arange(4000) + 0.5) / 4000.0                       # midpoint rule over the band
        for i in range(40):
            f = (low[i] + (high[i] - low[i]) * q)[:, None] / sr     # cycles per sample
            band = high[i] - low[i]
            hc = (2.0 * np.cos(2 * np.pi * f * n[None, :])).mean(0) * band          # = the integral over Hz
            hs = (2.0 * np.sin(2 * np.pi * f * n[None, :])).mean(0) * band
            want[i] = hc * win / (2.0 * band)
            want[40 + i] = hs * win / (2.0 * band)
        got_oracle = fb.filters().detach().numpy()[:, 0, :].astype(np.float64)
        got_product = sinc_filters(fb.low_hz_.detach(), fb.band_hz_.detach(), fb.window_, fb.n_).numpy().reshape(80, ks).astype(np.float64)
        scale = np.abs(want).max(axis=1, keepdims=True)
        assert (np.abs(got_oracle - want) / scale).max() < 2e-4, (np.abs(got_oracle - want) / scale).max()
        assert (np.abs(got_product - want) / scale).max() < 2e-4
        assert np.array_equal(got_oracle[:, ::-1][:40], got_oracle[:40])            # even
        assert np.array_equal(-got_oracle[:, ::-1][40:], got_oracle[40:])           # odd
        # frequency response: largest inside the band, >= 20 dB down two main-lobe widths outside it
        H = np.abs(np.fft.rfft(got_oracle, 8192, axis=1))
        fr = np.fft.rfftfreq(8192, 1 / sr)
        lobe = 2.0 * sr / ks
        for i in range(80):
            lo_, hi_ = low[i % 40], high[i % 40]
            inside = (fr >= lo_) & (fr <= hi_)
            outside = (fr < lo_ - 2 * lobe) | (fr > hi_ + 2 * lobe)
            if hi_ - lo_ < 30.0 or not inside.any() or not outside.any():
                continue
            assert fr[np.argmax(H[i])] >= lo_ - lobe and fr[np.argmax(H[i])] <= hi_ + lobe
            assert H[i][outside].max() < 0.1 * H[i][inside].max()
