"""dl_vqa_amd — the VqaNet train-step hot path of OmerShubi/DL_VQA on MI355X (gfx950).

``VqaNet`` is a drop-in ``torch.nn.Module`` (reference models/model.py:7-67) whose forward and
backward run on hand-written HIP kernels (csrc/, bound through the C ABI in include/vqa_hip.h);
``train`` mirrors the reference training procedure (train.py) with a fused device-side loss and
Adam; ``distributed.DataParallel`` shards the minibatch over the GPUs of one node with RCCL.
"""
from .model import VqaNet, questionNet, ImageNet2, Attention, Classifier  # noqa: F401

__all__ = ["VqaNet", "questionNet", "ImageNet2", "Attention", "Classifier"]
