"""NSF cfg2 Adam steps for a kernel trace (scripts/train_launches.sh): STEPS identical steps after one-time setup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import MAF, NSF

dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = (MAF(64, 0, transforms=8, hidden_features=[256] * 3) if os.environ.get("FLOW") == "maf" else NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3)).to(dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
x = torch.randn(1 << int(os.environ.get("LOG2B", "16")), 64, device=dev)
for _ in range(int(os.environ.get("STEPS", "12"))):
    loss = -flow().log_prob(x).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("loss", float(loss))
