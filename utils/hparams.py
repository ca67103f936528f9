from neuralsvb_amd.utils.hparams import hparams, set_hparams  # noqa: F401  (same dict object)
