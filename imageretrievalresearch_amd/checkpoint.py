"""Checkpoint ingest for the hot path: the ``load_checkpoint`` of inference/inference.py:77-149.

The reference's function does not parse (SyntaxError at :77: non-default argument after defaults); this restates its
intent with the same argument names and the same key handling:

* Lightning checkpoints: ``checkpoint['state_dict']`` keys carry the ``model.`` prefix of the LightningModule
  attribute (train/train.py:136); every occurrence of ``"model."`` is removed (``k.replace("model.", "")``,
  inference/inference.py:117-121) and the result is loaded with ``strict=False`` (:124).
* ``conv_input=True`` builds ``Sequential(conv_layer, base_model)`` (:101-105), whose keys are ``0.0.weight`` and
  ``1.<timm key>``.
* plain torch checkpoints: ``state_dict['state_dict']`` loaded strictly, then the classifier is replaced (:133-141).

Files are read with ``torch.load(..., weights_only=True)`` only: nothing from a checkpoint is ever executed.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import models


def strip_lightning_prefix(state_dict) -> "OrderedDict[str, torch.Tensor]":
    """inference/inference.py:113-121."""
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k.replace("model.", "")] = v
    return out


def load_checkpoint(checkpoint_path, model_name, pretrained=False, num_classes=0, from_pytorch_lightning=True,
                    conv_input=True, device="cuda:0"):
    """Build the model for ``model_name`` and load ``checkpoint_path`` into it; returns the model
    (not yet moved to ``device``, exactly like the reference: ``inference()`` calls ``model.to(device)``)."""
    if from_pytorch_lightning:
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        if conv_input:
            base_model = models.create_model(model_name)                      # :102 (default 1000-way head kept)
            model = models.with_conv_input(base_model)                        # :103-105
        else:
            model = models.create_model(model_name, num_classes=num_classes)  # :110
        new_state_dict = strip_lightning_prefix(checkpoint["state_dict"])
        result = model.load_state_dict(new_state_dict, strict=False)          # :124
        model.load_report = result
        return model
    if pretrained:
        model = models.create_model(model_name)                               # :133
        state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        model.load_state_dict(state_dict["state_dict"])                       # :137
        head = model.head.fc if hasattr(model, "head") and hasattr(model.head, "fc") else None
        num_features = head.in_features if isinstance(head, torch.nn.Linear) else model.num_features   # :140
        new_head = torch.nn.Linear(num_features, num_classes) if num_classes > 0 else torch.nn.Identity()
        if model.family == "efficientnet":
            model.classifier = new_head                                       # :141
        elif model.family == "rexnet":
            model.head.fc = new_head
            model.mark_dirty()
        else:
            model.head = new_head
        return model
    return models.create_model(model_name, num_classes=num_classes)           # :146
