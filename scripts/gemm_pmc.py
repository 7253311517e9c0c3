import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incremental_multimodal_medical_learning_ii_amd import kernels as K
dev = "cuda"; T = 32768
x768 = torch.randn(T, 768, device=dev); w1 = torch.randn(3072, 768, device=dev); y = torch.empty(T, 3072, device=dev)
for _ in range(3): K.linear_fwd(x768, w1, out=y)
N=256; H=28; C=128; Ko=128
x = torch.randn(N, H, H, C, device=dev); w = torch.randn(Ko, 3, 3, C, device=dev) * 0.05; sh = torch.zeros(Ko, device=dev); yy = torch.empty(N, H, H, Ko, device=dev)
for _ in range(3): K.conv_fwd(x, w, sh, None, yy, N, H, H, C, Ko, 3, 3, 1, 1, True)
torch.cuda.synchronize()
