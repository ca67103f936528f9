"""debug: eager vs hip_graph step logs (GPU only)"""
import os, sys, tempfile, pathlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_task_step as T
dev = torch.device("cuda:0")
res = {}
warm = int(os.environ.get("WARM", "1"))
for mode in ("eager", "graph"):
    tmp = pathlib.Path(tempfile.mkdtemp())
    task, trainer, batch, hp = T._setup(tmp, dev)
    trainer.hip_graph, trainer.hip_graph_warmup = mode == "graph", warm
    for m in task.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    g = torch.Generator().manual_seed(3)
    L = hp["latent_size"]
    eps_a, eps_p = torch.randn(2, L, 1, generator=g).to(dev), torch.randn(2, L, 1, generator=g).to(dev)
    orig_run = task.run_model
    task.run_model = lambda *a, _o=orig_run, **k: _o(*a, eps_a2a=eps_a, eps_p2p=eps_p, **k)
    host_lens = {k: batch[k].cpu() for k in ("mel_lengths", "prof_mel_lengths")}
    logs = []
    for step in range(1, 6):
        np.random.seed(200 + step)
        task.global_step = trainer.global_step = step
        pbar, _ = trainer.run_training_batch(0, dict(batch, **host_lens))
        torch.cuda.synchronize()
        logs.append({k: float(v) for k, v in pbar.items() if isinstance(v, torch.Tensor)})
    res[mode] = logs
for i, (a, b) in enumerate(zip(res["eager"], res["graph"])):
    print("step", i + 1)
    for k in a:
        flag = "" if abs(a[k] - b[k]) <= 1e-5 * max(1, abs(a[k])) else "   <<<<"
        print(f"   {k:10s} {a[k]:12.6f} {b[k]:12.6f}{flag}")
