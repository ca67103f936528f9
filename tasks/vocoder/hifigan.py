"""`tasks.vocoder.hifigan.HifiGanTask` (named by the reference's hifigan.yaml:2, absent upstream) -> MI355X task."""
from neuralsvb_amd.tasks.hifigan_task import HifiGanTask  # noqa: F401
