import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import functional as K
E, D = 544230, 200
x = torch.randn(E, D, device="cuda"); W = torch.randn(D, D, device="cuda") / 14; b = torch.randn(D, device="cuda")
for _ in range(5):
    with torch.no_grad():
        y = K.linear(x, W, b, "relu")
torch.cuda.synchronize()
