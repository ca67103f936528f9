from neuralsvb_amd.vocoders.base_vocoder import BaseVocoder, VOCODERS, get_vocoder_cls, register_vocoder  # noqa: F401
