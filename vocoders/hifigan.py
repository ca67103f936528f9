from neuralsvb_amd.vocoders.hifigan import HifiGAN, load_model  # noqa: F401
