"""Reference-compatible import paths (tasks.*) -> neuralsvb_amd.tasks.*"""
