"""Reference-compatible import path (vocoders.*) -> neuralsvb_amd.vocoders.*"""
from neuralsvb_amd.vocoders import hifigan  # noqa: F401
