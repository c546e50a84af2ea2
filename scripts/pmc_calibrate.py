"""Known-size HBM traffic for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box:
a 1 GiB device-to-device copy (reads 1 GiB, writes 1 GiB; larger than the 256 MiB Infinity Cache)."""
import torch

n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(3)
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
print("copied", n, "bytes x3")
