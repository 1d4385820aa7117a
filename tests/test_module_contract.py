"""The forward-only contract of the timm-shaped Module, CPU only: what the reference's TRAINING callers
(train/train.py:136,194-195) would hit must fail loudly instead of silently training nothing, and replacing a tensor
object anywhere in the tree must invalidate the packed HIP weights."""
import pytest
import torch
import torch.nn as nn

import imageretrievalresearch_amd as M


def test_train_mode_with_grad_is_refused_before_anything_runs():
    model = M.create_model("efficientnet_b3a", num_classes=0)
    assert model.training                                   # nn.Module default, as timm's
    x = torch.zeros(1, 3, 224, 224)                         # a CPU tensor: the guard must fire before the device check
    for call in (model, model.forward_features, model.embed):
        with pytest.raises(M.MI355Error, match="forward-only"):
            call(x)
    with pytest.raises(M.MI355Error, match="forward-only"):
        model.forward_uint8(torch.zeros(1, 224, 224, 3, dtype=torch.uint8))
    # eval mode, or train mode under no_grad (Lightning's validation loop), pass the guard and reach the device check
    model.eval()
    with pytest.raises(M.MI355Error, match="GPU|cuda|device"):
        model(x)
    model.train()
    with torch.no_grad(), pytest.raises(M.MI355Error, match="GPU|cuda|device"):
        model(x)


def test_classifier_head_refuses_to_drop_autograd_silently():
    model = M.create_model("rexnet_150", num_classes=7)
    fm = torch.zeros(2, 1920, 7, 7)
    model.train()                                           # train/train.py:194-195: lbl = self.model.head(fm)
    with pytest.raises(M.MI355Error, match="forward-only"):
        model.head(fm)
    model.eval()
    with pytest.raises(M.MI355Error, match="forward-only"):
        model.head(fm.clone().requires_grad_(True))         # a map that carries history, eval mode or not
    with pytest.raises(M.MI355Error, match="GPU|cuda|device"):
        model.head(fm)                                      # plain inference: passes the guard, then needs the GPU


def test_replacing_a_parameter_object_on_a_submodule_invalidates_the_pack():
    """`model.conv_stem.weight = nn.Parameter(...)` (or parametrize / prune) used to leave the cached tensor
    list watching the old tensor, so the signature never changed and the packed weights went stale."""
    model = M.create_model("efficientnet_b3a", num_classes=0).eval()
    model.__dict__["_dirty"] = False                        # as if packed
    model.__dict__["_sig"] = model._signature()
    assert model.__dict__["_sig_tensors"] is not None
    model.blocks[3][2].conv_pwl.weight = nn.Parameter(torch.ones_like(model.blocks[3][2].conv_pwl.weight))
    assert model._dirty and model.__dict__["_sig_tensors"] is None
    assert model._sig != model._signature()
    model.__dict__["_dirty"] = False
    model.__dict__["_sig"] = model._signature()
    model.bn1.register_buffer("running_mean", torch.ones_like(model.bn1.running_mean))   # re-registering a buffer too
    assert model._dirty and model._sig != model._signature()
    # in-place writes keep being seen through the version counters
    model.__dict__["_dirty"] = False
    model.__dict__["_sig"] = model._signature()
    with torch.no_grad():
        model.conv_head.weight.add_(1.0)
    assert model._sig != model._signature()
