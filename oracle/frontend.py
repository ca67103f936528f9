"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the mel / pitch front-end.

Not product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this.  The product path is neuralsvb_amd (HIP) and never routes through here.

PARITY UNPINNED for the librosa arithmetic: the reference calls librosa 0.8.0
(Requirements.txt:41), which is not vendored under /root/reference and is not
installed here, and the reference ships no golden vectors.  `librosa_stft` and
`librosa_mel_filterbank` restate the *published* librosa-0.8.0 algorithm
 and are cross-checked against torch.stft (librosa's conventions) and
the filterbank example of librosa's own documentation by
tests/test_oracle_golden.py::test_frontend_restatement_against_independent_implementations
-- anchors, not librosa's source: the header stays "unpinned"; everything downstream of them (log10/ln, eps, trimming,
pitch-bin quantisation) follows in-tree reference code and is pinned by reading, with
`f0_to_coarse` additionally pinned against the imported reference function
(tests/golden/make_golden.py).

Reference call sites restated here:
  data_gen/tts/data_gen_utils.py:93-147   process_utterance      -> wav2mel_offline
  utils/audio.py:67-76                    librosa_pad_lr         -> (inside wav2mel_offline)
  modules/hifigan/mel_utils.py:45-79      mel_spectrogram        -> mel_spectrogram_ingraph
  utils/pitch_utils.py:130-146            f0_to_coarse           -> f0_to_coarse
  utils/pitch_utils.py:149-195            norm_f0/norm_interp_f0/denorm_f0
  data_gen/tts/data_gen_utils.py:150-184  get_pitch (alignment arithmetic only) -> align_f0_to_mel
"""
import numpy as np


# ----------------------------------------------------------------------------------------------
# librosa 0.8.0 restatements (third-party arithmetic; published algorithm)
# ----------------------------------------------------------------------------------------------
def hann_periodic(win_length: int) -> np.ndarray:
    """scipy.signal.get_window('hann', N, fftbins=True): 0.5 - 0.5 cos(2 pi n / N)."""
    n = np.arange(win_length, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length))


def librosa_stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
                 pad_mode="reflect", **_):
    """librosa.stft (0.8.0 semantics): [1+n_fft/2, 1+len(y)//hop] complex64."""
    y = np.asarray(y)
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 4 if hop_length is None else hop_length
    assert window == "hann"
    w = hann_periodic(win_length)
    if win_length < n_fft:  # librosa.util.pad_center
        lp = (n_fft - win_length) // 2
        w = np.pad(w, (lp, n_fft - win_length - lp))
    w = w.astype(np.float32)
    if center:
        mode = "constant" if pad_mode == "constant" else pad_mode
        y = np.pad(y, n_fft // 2, mode=mode)
    n_frames = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    frames = y[idx].astype(np.float32) * w[None, :]
    spec = np.fft.rfft(frames, axis=1).astype(np.complex64)
    return spec.T  # [bins, frames]


def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def librosa_mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **_):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with htk=False, norm='slaney' -> float32 [n_mels, 1+n_fft/2]."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


# ----------------------------------------------------------------------------------------------
# in-tree reference arithmetic
# ----------------------------------------------------------------------------------------------
def wav2mel_offline(wav, fft_size=512, hop_size=128, win_length=512, num_mels=80, fmin=50, fmax=11025,
                    sample_rate=22050, eps=1e-10):
    """data_gen_utils.py:93-147 (`process_utterance`, vocoder='pwg') with eps from vocoders/pwg.py:118.

    Returns (wav_trimmed[T*hop], mel[T, num_mels] float32 log10)."""
    wav = np.asarray(wav, dtype=np.float32)
    spc = np.abs(librosa_stft(wav, n_fft=fft_size, hop_length=hop_size, win_length=win_length,
                              pad_mode="constant"))                       # :123-125
    fmin = 0 if fmin == -1 else fmin
    fmax = sample_rate / 2 if fmax == -1 else fmax
    basis = librosa_mel_filterbank(sample_rate, fft_size, num_mels, fmin, fmax)  # :130
    mel = basis @ spc                                                     # :131
    mel = np.log10(np.maximum(eps, mel))                                  # :134
    r_pad = (wav.shape[0] // hop_size + 1) * hop_size - wav.shape[0]      # utils/audio.py:72
    wav = np.pad(wav, (0, r_pad))                                         # :138-139
    wav = wav[:mel.shape[1] * hop_size]                                   # :140
    return wav, mel.T.astype(np.float32)


def mel_spectrogram_ingraph(y, fft_size=512, hop_size=128, win_size=512, num_mels=80, fmin=50, fmax=11025,
                            sample_rate=22050):
    """modules/hifigan/mel_utils.py:45-79 restated with return_complex=True (the reference call
    raises on torch>=2).  y: torch [B, N] -> [B, num_mels, N//hop] (natural log)."""
    import torch
    y = y.clamp(min=-1.0, max=1.0)                                        # :58
    basis = torch.from_numpy(librosa_mel_filterbank(sample_rate, fft_size, num_mels, fmin, fmax)).to(y)
    win = torch.hann_window(win_size, periodic=True).to(y)                 # :64
    p = int((fft_size - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)   # :66-68
    spec = torch.stft(y, fft_size, hop_length=hop_size, win_length=win_size, window=win, center=False,
                      normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)                          # :74
    spec = torch.matmul(basis, spec)                                       # :75
    return torch.log(torch.clamp(spec, min=1e-5))                          # :76, :23


F0_BIN = 256
F0_MAX = 1100.0
F0_MIN = 50.0
F0_MEL_MIN = 1127 * np.log(1 + F0_MIN / 700)
F0_MEL_MAX = 1127 * np.log(1 + F0_MAX / 700)


def f0_to_coarse(f0):
    """utils/pitch_utils.py:130-146.  numpy input -> np.rint (half-to-even); torch input -> (x+0.5).long()."""
    import torch
    is_torch = isinstance(f0, torch.Tensor)
    if is_torch:
        mel = 1127 * (1 + f0 / 700).log()
    else:
        mel = 1127 * np.log(1 + np.asarray(f0) / 700)
    pos = mel > 0
    mel = mel.clone() if is_torch else mel.copy()
    mel[pos] = (mel[pos] - F0_MEL_MIN) * (F0_BIN - 2) / (F0_MEL_MAX - F0_MEL_MIN) + 1
    mel[mel <= 1] = 1
    mel[mel > F0_BIN - 1] = F0_BIN - 1
    return (mel + 0.5).long() if is_torch else np.rint(mel).astype(np.int64)


def norm_interp_f0(f0, f0_mean, f0_std, pitch_norm="standard", use_uv=True):
    """utils/pitch_utils.py:149-176 for numpy input: standardise, zero unvoiced, linear-interp unvoiced."""
    f0 = np.asarray(f0, dtype=np.float64).copy()
    uv = f0 == 0
    if pitch_norm == "standard":
        f0 = (f0 - f0_mean) / f0_std
    elif pitch_norm == "log":
        f0 = np.log2(f0 + 1e-8)
    if use_uv:
        f0[uv] = 0
    if uv.sum() == len(f0):
        f0[uv] = 0
    elif uv.sum() > 0:
        f0[uv] = np.interp(np.where(uv)[0], np.where(~uv)[0], f0[~uv])
    return f0, uv


def denorm_f0(f0, uv, f0_mean, f0_std, pitch_norm="standard", use_uv=True, f0_max=F0_MAX):
    """utils/pitch_utils.py:179-195 (torch tensors)."""
    if pitch_norm == "standard":
        f0 = f0 * f0_std + f0_mean
    elif pitch_norm == "log":
        f0 = 2 ** f0
    f0 = f0.clamp(min=0).clamp(max=f0_max)
    if uv is not None and use_uv:
        f0 = f0.clone()
        f0[uv > 0] = 0
    return f0


def align_f0_to_mel(f0_raw, n_mel_frames, hop_size=128):
    """data_gen_utils.py:160-183: left-pad by 2*pad_size frames, right pad / trim to len(mel)."""
    pad_size = {128: 4, 256: 2}[hop_size]
    lpad = pad_size * 2
    rpad = n_mel_frames - len(f0_raw) - lpad
    if rpad < 0:
        raise ValueError("np.pad would raise on negative rpad in the reference (data_gen_utils.py:173)")
    f0 = np.pad(np.asarray(f0_raw, dtype=np.float64), [[lpad, rpad]], mode="constant")
    delta_l = n_mel_frames - len(f0)
    assert abs(delta_l) <= 8
    if delta_l > 0:
        f0 = np.concatenate([f0, [f0[-1]] * delta_l], 0)
    f0 = f0[:n_mel_frames]
    return f0, f0_to_coarse(f0)
