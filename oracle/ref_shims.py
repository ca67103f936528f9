"""TEST INFRASTRUCTURE ONLY -- import shims for running the *unmodified* reference.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.

This module makes ``/root/reference`` importable in the build container (SURVEY.md
Appendix E): the reference imports ~10 third-party packages that are not installed
(chardet, librosa, h5py, pycwt, parselmouth, webrtcvad, pyloudnorm, resemblyzer,
skimage, tensorboard) and uses numpy aliases removed in numpy 2 (np.int, np.float ...).
We insert *stub modules* for the former and restore the latter.  No reference source
is copied; the reference tree is read where it lies and is only available in the build
container (never on the GPU box), so this is used exclusively by
``tests/golden/make_golden.py`` to generate the committed golden vectors and by
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent).

``librosa.stft`` / ``librosa.filters.mel`` are routed to the numpy restatements in
``oracle/frontend.py`` (librosa 0.8.0 is pinned by the reference's Requirements.txt:41
but is not vendored: "parity unpinned" for that arithmetic -- see DESIGN.md).
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("SVB_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modules", "voice_conversion"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so "import a.b" works
    sys.modules[name] = m
    return m


class _Anything:
    """Callable/attribute sink used for third-party symbols the hot path never executes."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, item):
        return _Anything()


class _NullSummaryWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_audio(self, *a, **k):
        pass

    def add_figure(self, *a, **k):
        pass

    def close(self):
        pass


def install(chdir_to_reference: bool = False):
    """Install stubs + numpy aliases and put the reference on sys.path.  Idempotent."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for alias, typ in (("int", int), ("float", float), ("bool", bool), ("complex", complex),
                       ("object", object)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)
    if "Inf" not in np.__dict__:
        np.Inf = np.inf

    from oracle import frontend as _fe

    if "chardet" not in sys.modules:
        _stub("chardet", detect=lambda b: {"encoding": "utf-8"})
    if "librosa" not in sys.modules:
        lib = _stub("librosa")
        filt = _stub("librosa.filters", mel=_fe.librosa_mel_filterbank)
        core = _stub("librosa.core", load=_Anything())
        lib.filters = filt
        lib.core = core
        lib.stft = _fe.librosa_stft
        lib.istft = _Anything()
        lib.piptrack = _Anything()
        lib.effects = _Anything()
        _stub("librosa.util", normalize=_Anything())
    for name in ("h5py", "pycwt", "pycwt.wavelet", "parselmouth", "webrtcvad", "pyloudnorm",
                 "resemblyzer", "skimage", "skimage.transform", "numba", "g2p_en", "pypinyin",
                 "jieba", "soundfile", "tensorboard"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["resemblyzer"].VoiceEncoder = _Anything
    sys.modules["skimage.transform"].resize = _Anything()
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    sys.modules["pycwt"].wavelet = sys.modules["pycwt.wavelet"]
    def _jit(*a, **k):               # both `@jit` and `@jit(...)` leave the function as plain Python
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    sys.modules["numba"].jit = _jit
    sys.modules["h5py"].File = _Anything
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        _stub("torch.utils.tensorboard", SummaryWriter=_NullSummaryWriter)
        import torch.utils
        torch.utils.tensorboard = sys.modules["torch.utils.tensorboard"]

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # The reference's `tasks/` has no __init__.py (namespace package); the product keeps a regular `tasks` package at the
    # repo root for the drop-in dotted paths, and regular packages win over namespace packages whatever sys.path says.
    # Pin the reference's directories explicitly for this (checker) process.
    for top in ("tasks", "vocoders", "modules", "data_gen"):
        ref_dir = os.path.join(REFERENCE_ROOT, top)
        cur = sys.modules.get(top)
        if os.path.isdir(ref_dir) and not os.path.exists(os.path.join(ref_dir, "__init__.py")) and \
                (cur is None or ref_dir not in list(getattr(cur, "__path__", []))):
            for k in [k for k in sys.modules if k == top or k.startswith(top + ".")]:
                del sys.modules[k]
            m = types.ModuleType(top)
            m.__path__ = [ref_dir]
            sys.modules[top] = m
    if chdir_to_reference:
        os.chdir(REFERENCE_ROOT)


def set_reference_hparams(overrides=None):
    """Resolve the reference's own YAML chain for vae_global_mle_eng and apply overrides.

    YAML base_config entries are cwd-relative (reference utils/hparams.py:59-61), so the
    chain is resolved with cwd = reference root and restored afterwards.
    """
    install()
    from utils.hparams import set_hparams, hparams  # reference module
    cwd = os.getcwd()
    try:
        os.chdir(REFERENCE_ROOT)
        set_hparams(config="egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml", exp_name="",
                    print_hparams=False)
    finally:
        os.chdir(cwd)
    if overrides:
        hparams.update(overrides)
    return hparams
