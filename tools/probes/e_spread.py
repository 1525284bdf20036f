"""How much does E move between seeds / halves of a batch at config 2?  (E_loc is heavy-tailed: Coulomb, no cusp in the flow.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as G
dev = torch.device("cuda:0")
model = G._model(dev, 3, 3, 2.0)
for seed in (1234, 1, 2, 3):
    for B in (65536, 131072):
        torch.manual_seed(seed)
        model(B)
        e = model.Eloc
        h = B // 2
        srt = torch.sort(e).values
        print(f"seed {seed} B {B}: E {model.E:.4f} std {model.E_std:.3f} | halves {e[:h].mean().item():.4f} {e[h:].mean().item():.4f} | "
              f"min {srt[0].item():.1f} max {srt[-1].item():.1f} | trimmed mean (0.1%) {srt[B//1000:-B//1000].mean().item():.4f} nan {torch.isnan(e).sum().item()}")
