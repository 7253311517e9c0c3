#!/usr/bin/env python3
"""Convert an adapter checkpoint written by the REFERENCE's `Trainer.save()` into the tensors-only file this package loads.

The reference pickles whole modules (`torch.save(self.image_adapter, log_dir + '/image_adapter.pt')`, Trainer.py:1643-1648):
loading such a file executes pickle code and needs the reference's `models` module on the import path.  This package only opens
files with `torch.load(..., weights_only=True)` and stores `module.state_dict()` under the same file names.

Run this ONCE, in the reference's own environment (its repository root on sys.path, so that `models.myMLP` unpickles), on files
you trust; it is never imported or executed by the package, its tests or bench.py:

    python convert_reference_adapter.py NUOVI_RISULTATI/.../image_adapter.pt converted/image_adapter.pt
"""
import sys

import torch


def main(src: str, dst: str) -> None:
    module = torch.load(src, map_location="cpu")          # whole-module pickle: reference environment only
    if not isinstance(module, torch.nn.Module):
        raise SystemExit(f"{src}: expected a pickled nn.Module (models.myMLP / models.myLinearModel), got {type(module).__name__}")
    sd = {k: v.detach().cpu().contiguous() for k, v in module.state_dict().items()}
    torch.save(sd, dst)
    print(f"{dst}: {len(sd)} tensors: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in sd.items()))


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2])
