"""Golden vectors for the photometric MSE loss from the REFERENCE's own ``LossMse`` (build container only).

/root/reference/src/loss/loss_mse.py is imported under a synthetic ``src.*`` package tree; its type-only imports
(jaxtyping annotations, dataset / decoder / Gaussians types) get empty stand-in modules -- the class itself and
``Loss.__init__`` (loss.py) run unmodified.  Writes tests/golden/loss_goldens.pt.   python tests/golden/make_loss_goldens.py
"""
import importlib.util
import sys
import types
from pathlib import Path

import torch

REF = Path("/root/reference/src")
HERE = Path(__file__).resolve().parent


class _Ann:
    def __class_getitem__(cls, item):
        return cls


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference_loss():
    _module("jaxtyping", Float=type("Float", (_Ann,), {}))
    for p in ("src", "src.loss", "src.dataset", "src.model", "src.model.decoder"):
        _module(p)
    _module("src.dataset.types", BatchedExample=dict)
    _module("src.model.decoder.decoder", DecoderOutput=object)
    _module("src.model.types", Gaussians=object)
    for name in ("loss", "loss_mse"):
        spec = importlib.util.spec_from_file_location(f"src.loss.{name}", REF / "loss" / f"{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"src.loss.{name}"] = m
        spec.loader.exec_module(m)
    return sys.modules["src.loss.loss_mse"]


def main():
    ref = load_reference_loss()
    gen = torch.Generator().manual_seed(11)
    out = {}
    for name, (shape, weight, after, step) in {
        "b2v3_32x40_w1": ((2, 3, 3, 32, 40), 1.0, 0, 10),
        "b1v2_17x13_w05": ((1, 2, 3, 17, 13), 0.5, 0, 0),          # numel not a multiple of 4
        "b1v1_64x64_w2_late": ((1, 1, 3, 64, 64), 2.0, 100, 100),
        "b1v1_8x8_not_yet": ((1, 1, 3, 8, 8), 1.0, 100, 99),        # before apply_after_step -> 0
    }.items():
        pred = torch.rand(shape, generator=gen).requires_grad_(True)
        img = torch.rand(shape, generator=gen)
        loss_mod = ref.LossMse(ref.LossMseCfgWrapper(ref.LossMseCfg(weight=weight, apply_after_step=after)))
        assert loss_mod.name == "mse"
        loss = loss_mod(pred, img, None, step)
        grad = torch.zeros_like(pred)
        if loss.requires_grad:
            (grad,) = torch.autograd.grad(loss, pred)
        out[name] = {"prediction": pred.detach(), "image": img, "weight": weight, "apply_after_step": after,
                     "global_step": step, "loss": loss.detach(), "grad": grad}
        print(name, float(loss))
    torch.save(out, HERE / "loss_goldens.pt")
    print("wrote", HERE / "loss_goldens.pt")


if __name__ == "__main__":
    main()
