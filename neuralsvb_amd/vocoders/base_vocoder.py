"""Vocoder plugin registry (reference vocoders/base_vocoder.py:5-39): `hparams['vocoder']` is a registered name
(case-insensitive) or a dotted `pkg.Class` path."""
import importlib

VOCODERS = {}


def register_vocoder(cls):
    VOCODERS[cls.__name__.lower()] = cls
    VOCODERS[cls.__name__] = cls
    return cls


def get_vocoder_cls(hparams):
    name = hparams["vocoder"]
    if name in VOCODERS:
        return VOCODERS[name]
    if name.lower() in VOCODERS:
        return VOCODERS[name.lower()]
    pkg, cls_name = name.rsplit(".", 1)
    return getattr(importlib.import_module(pkg), cls_name)


class BaseVocoder:
    def spec2wav(self, mel, **kwargs):
        """mel [T, 80] -> wav [T*hop] (np.float32)"""
        raise NotImplementedError

    @staticmethod
    def wav2spec(wav_fn):
        """wav file -> (wav [T*hop], mel [T, 80])"""
        raise NotImplementedError
