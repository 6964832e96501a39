"""tools/pmc_calib.py -- known-size traffic for calibrating FETCH_SIZE / WRITE_SIZE: fill 2 GiB, copy 2 GiB (5 times each)."""
import torch
n = 2 * 2 ** 30 // 4
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(5):
    a.fill_(1.5)
torch.cuda.synchronize()
for _ in range(5):
    b.copy_(a)
torch.cuda.synchronize()
