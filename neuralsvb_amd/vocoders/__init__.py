from . import hifigan  # noqa: F401  (registers "HifiGAN"/"hifigan")
