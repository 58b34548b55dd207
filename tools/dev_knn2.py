import sys, torch
sys.path.insert(0, ".")
from samble_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
x = torch.randn(32, 128, 2048, generator=torch.Generator().manual_seed(0)).to(dev)
lib.samble_knn_tri_config(1, 2)
for _ in range(5):
    ops.stage_knn(x, x, 32)
torch.cuda.synchronize()
