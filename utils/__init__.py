"""Reference-compatible import paths (utils.*) -> neuralsvb_amd.utils.*"""
