"""`tasks.singing.svb_vae_task` of the reference -> the MI355X task (drop-in dotted path for `task_cls`)."""
from neuralsvb_amd.tasks.dataset import MultiSpkEmbDataset  # noqa: F401
from neuralsvb_amd.tasks.svb_vae_task import SVBVAEMleTask  # noqa: F401
