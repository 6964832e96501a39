import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
for i in range(2):
    out = bench.consumer_rows(mesh, model, dev, 800, 800)
    for k, v in out.items():
        print(i, k[:40], {a[:30]: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a != "steps"})
