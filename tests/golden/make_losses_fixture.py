#!/usr/bin/env python3
"""Generates tests/golden/F13_losses.npz by EXECUTING the reference's own loss and score code in this container (CPU; float32 — the
reference's losses build float32 one-hot / target tensors (lovasz_loss.py:46, diceloss.py:190), so they only run in float32):

  /root/reference/latticenet_py/lattice/lovasz_loss.py:23   LovaszSoftmax (ln_train.py:128-157 uses it next to NLL)
  /root/reference/latticenet_py/lattice/diceloss.py:8       GeneralizedSoftDiceLoss
  /root/reference/latticenet_py/callbacks/scores.py:8-110   Scores (accumulate_scores, compute_stats, update_best)

Stand-ins for what the image lacks: `torchnet` (imported by scores.py, never called by the code paths used) is an empty module;
"cuda" maps to the CPU (Tensor.to / torch.ones(..).to("cuda"), torch.cuda.FloatTensor -> a CPU float tensor), as in
make_reference_network_fixture.py.  Nothing of the reference is patched.

What is written is DATA ONLY: per case the seeded inputs (log-probabilities [N, C] float64 — cast to float32 for the reference —, labels [N]), the reference's loss values
and their gradients with respect to the log-probabilities, and for Scores the per-class IoU / average after each accumulated cloud.
Cases: plain, an ignore class, classes absent from the cloud, every point of one class, one point, two clouds accumulated.

Run in the build container only:  python tests/golden/make_losses_fixture.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/latticenet_py"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_stand_ins():
    sys.modules.setdefault("torchnet", types.ModuleType("torchnet"))
    orig_to = torch.Tensor.to

    def to(self, *a, **k):  # "cuda" -> where the tensor is
        a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
        k = {kk: vv for kk, vv in k.items() if not (kk == "device" and isinstance(vv, str) and vv.startswith("cuda"))}
        return orig_to(self, *a, **k) if (a or k) else self

    torch.Tensor.to = to
    torch.cuda.FloatTensor = lambda *shape: torch.zeros(*shape, dtype=torch.float32)  # diceloss.py:190 (filled with 0 there)


def cases():
    """(name, n, c, ignore, label generator)"""
    rng = np.random.default_rng(1234)
    out = []

    def case(name, n, c, ignore, labels, sharp=1.0):
        logits = rng.standard_normal((n, c)) * sharp
        out.append((name, torch.log_softmax(torch.from_numpy(logits), 1).numpy(), np.asarray(labels, np.int64), ignore))

    case("plain", 500, 6, 0, rng.integers(0, 6, 500))
    case("absent_classes", 400, 8, 0, rng.integers(1, 4, 400))              # classes 4..7 never occur (and 0 is ignored)
    case("ignore_last", 300, 5, 4, rng.integers(0, 5, 300))
    case("one_class", 200, 4, 0, np.full(200, 2))                           # every point of one class
    case("one_point", 1, 3, 0, [1])
    case("confident", 600, 20, 0, rng.integers(0, 20, 600), sharp=6.0)      # the SemanticKITTI head's class count
    return out


def main():
    install_stand_ins()
    lov = load(os.path.join(REF, "lattice", "lovasz_loss.py"), "ref_lovasz_loss")
    dice = load(os.path.join(REF, "lattice", "diceloss.py"), "ref_diceloss")
    scores = load(os.path.join(REF, "callbacks", "scores.py"), "ref_scores")
    data = {}
    names = []
    for name, logp, labels, ignore in cases():
        names.append(name)
        data[f"{name}/logp"], data[f"{name}/labels"], data[f"{name}/ignore"] = logp, labels, np.int64(ignore)
        for red in ("mean", "sum"):
            x = torch.from_numpy(logp).float().requires_grad_(True)
            loss = lov.LovaszSoftmax(ignore_index=ignore, reduction=red)(x, torch.from_numpy(labels))
            loss.backward()
            data[f"{name}/lovasz_{red}"], data[f"{name}/lovasz_{red}_grad"] = np.float64(loss.item()), x.grad.numpy()
        x = torch.from_numpy(logp).float()
        data[f"{name}/lovasz_none"] = lov.LovaszSoftmax(ignore_index=ignore, reduction="none")(x, torch.from_numpy(labels)).numpy()
        x = torch.from_numpy(logp).float().requires_grad_(True)
        loss = dice.GeneralizedSoftDiceLoss(ignore_index=ignore)(x, torch.from_numpy(labels))
        loss.backward()
        data[f"{name}/dice"], data[f"{name}/dice_grad"] = np.float64(loss.item()), x.grad.numpy()
    # Scores: clouds accumulated one after the other (ln_eval / ln_train: accumulate_scores per cloud, stats at the end of the epoch)
    rng = np.random.default_rng(99)
    s = scores.Scores()
    c = 7
    for k in range(3):
        n = 300 + 50 * k
        gt = rng.integers(0, c - (1 if k == 0 else 0), n)        # class 6 absent from the first cloud
        probs = torch.softmax(torch.from_numpy(rng.standard_normal((n, c)) + 2.5 * np.eye(c)[gt]), 1)
        s.accumulate_scores(probs, torch.from_numpy(gt), 0)
        avg, d = s.compute_stats()
        data[f"scores/{k}/probs"], data[f"scores/{k}/gt"] = probs.numpy(), gt
        data[f"scores/{k}/avg_iou"] = np.float64(avg)
        data[f"scores/{k}/iou_classes"] = np.array(sorted(d), np.int64)
        data[f"scores/{k}/iou_values"] = np.array([d[i] for i in sorted(d)], np.float64)
        s.update_best()
        data[f"scores/{k}/best_iou"] = np.float64(s.best_iou)
    data["case_names"] = np.array(names)
    data["scores/unlabeled_idx"] = np.int64(0)
    out = os.path.join(HERE, "F13_losses.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out), "bytes;", ", ".join(f"{n}: lovasz {float(data[n + '/lovasz_mean']):.6f} dice {float(data[n + '/dice']):.6f}" for n in names))


if __name__ == "__main__":
    main()
