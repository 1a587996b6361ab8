"""Box calibration: what does this MI355X deliver on a library GEMM and a device copy (comparison only)."""
import torch, time
d = "cuda"
print(torch.cuda.get_device_name(0))
for n in (4096, 8192):
    a = torch.randn(n, n, device=d, dtype=torch.bfloat16); b = torch.randn(n, n, device=d, dtype=torch.bfloat16)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): c = a @ b
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print("gemm bf16 %d: %.1f TFLOP/s" % (n, 2 * n ** 3 / dt / 1e12))
x = torch.empty(1 << 30, device=d, dtype=torch.uint8); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t = time.time()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt = (time.time() - t) / 10
print("copy 1GiB: %.2f TB/s (read+write)" % (2 * (1 << 30) / dt / 1e12))
# conv via MIOpen for comparison: 3x3 512->1024 13x13 batch 32, and 3x3 128->256 52x52
import torch.nn.functional as F
for (n, c, h, k) in ((32, 512, 13, 1024), (32, 128, 52, 256), (32, 256, 26, 512), (32, 32, 208, 64)):
    xx = torch.randn(n, c, h, h, device=d, dtype=torch.bfloat16).to(memory_format=torch.channels_last)
    w = torch.randn(k, c, 3, 3, device=d, dtype=torch.bfloat16).to(memory_format=torch.channels_last)
    for _ in range(3): o = F.conv2d(xx, w, padding=1)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): o = F.conv2d(xx, w, padding=1)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print("miopen conv3x3 n%d c%d h%d k%d: %.3f ms %.1f TFLOP/s" % (n, c, h, k, dt * 1e3, 2 * 9 * c * k * h * h * n / dt / 1e12))
