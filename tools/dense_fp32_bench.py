import sys, time, torch
sys.path.insert(0, ".")
import torch.nn.functional as F
torch.backends.cudnn.benchmark = ("bench" in sys.argv)
x = torch.randn(1, 128, 180, 180, device="cuda")
w = torch.randn(128, 128, 3, 3, device="cuda")
x2 = torch.randn(1, 256, 90, 90, device="cuda"); w2 = torch.randn(256, 256, 3, 3, device="cuda")
x3 = torch.randn(1, 512, 180, 180, device="cuda"); w3 = torch.randn(64, 512, 3, 3, device="cuda")
x4 = torch.randn(1, 64, 180, 180, device="cuda"); w4 = torch.randn(384, 64, 3, 3, device="cuda")
for name, a, b in (("128x128@180", x, w), ("256x256@90", x2, w2), ("512->64@180", x3, w3), ("64->384@180", x4, w4)):
    for _ in range(3): F.conv2d(a, b, padding=1)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): F.conv2d(a, b, padding=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    fl = 2 * a.shape[2] * a.shape[3] * b.shape[0] * b.shape[1] * 9
    print(name, "%.1f us  %.1f TFLOP/s" % (dt * 1e6, fl / dt / 1e12))
