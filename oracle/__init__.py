"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's embed-then-rank hot path, used as the
checker by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``imageretrievalresearch_amd/`` imports this package: the product path is the
HIP library and fails loudly when it is missing.

Pinning status:
* rank + loss (``oracle.rank``): PINNED — checked against the reference's own
  ``utils/contrastive_loss.py::ContrastiveLoss`` imported from /root/reference and against
  ``torch.nn.CosineSimilarity`` + ``torch.topk`` called exactly as ``train/train.py:250-251``;
  vectors committed under ``tests/golden/`` by ``tests/golden/make_golden.py``.
* backbones (``oracle.effnet`` / ``oracle.rexnet`` / ``oracle.swin``): PARITY UNPINNED — the
  arithmetic lives in timm==0.4.12 (requirements.txt:164), which is neither vendored in the
  reference nor installed here, and the reference holds no test or fixture for it.  The
  restatement follows timm 0.4.12's published structure and is anchored by exact parameter
  counts (efficientnet_b3 12 233 232; rexnet_150/200 9.73 M / 16.37 M; swin_base 87.77 M — see each module).
* pre-processing (``oracle.preprocess``): the Resize of train/train.py:48 is PINNED against Pillow itself (the
  third-party library torchvision calls for PIL images; Pillow 12.2.0 is installed here) in
  ``tests/test_preprocess.py`` and by ``tests/golden/resize_golden.npz`` (``make_resize_golden.py``); SquarePad is
  checked against ``PIL.ImageOps.expand``; ToTensor / Normalize are single-rounding fp32 formulas, unpinned against
  torchvision (not installed).
* score booster (``oracle.rank.score_boost``): the three published formulas of utils/score_booster.py, also checked in
  double on sample points in the GPU test.
"""
