"""CPU: the host-side mirror of the reference interface (no kernels run here)."""
import numpy as np
import pytest
import torch

from tests.golden_util import Golden, full_cfg, tiny_cfg


def test_state_dict_contract_matches_reference():
    from dl_vqa_amd import VqaNet
    g = Golden("tiny_plus")
    m = VqaNet(tiny_cfg(g.meta), g.meta["V"])
    sd = m.state_dict()
    assert list(sd.keys()) == list(g.sd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(g.sd[k].shape), k
    m.load_state_dict(g.sd)                       # a reference checkpoint's model_state loads as is
    # attributes read by utils/main_utils.py:21-41 (get_model_string)
    for attr in ("text", "image", "attention", "classifier"):
        assert sum(p.numel() for p in getattr(m, attr).parameters()) > 0
    assert "VqaNet" in str(m)


def test_seeded_init_is_bitwise_the_reference_init():
    """torch.manual_seed(s); VqaNet(cfg, V) must give the reference's initial weights (same layer
    constructors in the same order): checked against checksums recorded from the reference."""
    from dl_vqa_amd import VqaNet
    g = Golden("full224_seed1")
    torch.manual_seed(g.meta["seed"])
    m = VqaNet(full_cfg(g.meta["A"]), g.meta["V"])
    sd = m.state_dict()
    names = [str(n) for n in g.raw["param_names"]]
    assert list(sd.keys()) == names
    for n, s_ref, a_ref in zip(names, g.raw["param_sum"], g.raw["param_abs"]):
        assert float(sd[n].double().sum()) == pytest.approx(s_ref, rel=1e-12, abs=1e-12), n
        assert float(sd[n].double().abs().sum()) == pytest.approx(a_ref, rel=1e-12), n


def test_flat_order_groups_are_contiguous():
    from dl_vqa_amd.model import _flat_order
    from dl_vqa_amd import VqaNet
    g = Golden("tiny_plus")
    m = VqaNet(tiny_cfg(g.meta), g.meta["V"])
    order = _flat_order([n for n, _ in m.named_parameters()])
    groups = [n.split(".")[0] for n in order]
    assert groups == sorted(groups, key=["classifier", "attention", "text", "image"].index)
    assert order[-1].startswith("image.conv0")    # produced last by backward


def test_unsupported_configs_fail_loudly():
    from dl_vqa_amd import VqaNet
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    cfg["image"]["num_channels"] = [3, 6, 16, 32]
    with pytest.raises(ValueError, match="multiples of 4"):
        VqaNet(cfg, 10)
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    m = VqaNet(cfg, 10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32), torch.ones(1, 3, dtype=torch.int64), torch.tensor([3]))
    # num_lstm_layers > 1: the reference constructs but its own forward raises (questionNet returns [B, layers*ndir*H] while
    # q_lin / lin1 are sized for ndir*H, models/model.py:36,166,174; "needs change of code if >1", config.yaml:55 -- run here:
    # RuntimeError "mat1 and mat2 shapes cannot be multiplied (3x64 and 32x24)").  The mirror refuses at construction.
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    cfg["text"]["num_lstm_layers"] = 2
    with pytest.raises(NotImplementedError, match="num_lstm_layers"):
        VqaNet(cfg, 10)
    # every kernel_size of the schema constructs (3: implicit-GEMM kernels, others: csrc/conv_generic.hip); the opt-in
    # compute modes are 3 x 3 only
    for ks in (1, 2, 5):
        cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+", kernel_size=ks))
        assert tuple(VqaNet(cfg, 10).image.conv0.weight.shape) == (8, 3, ks, ks)
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+", kernel_size=5))
    cfg["image"]["num_channels"] = [3, 64, 64, 128]
    with pytest.raises(ValueError, match="kernel_size"):
        VqaNet(cfg, 10, compute_dtype="bf16")


def test_lr_schedule():
    from dl_vqa_amd.train import update_learning_rate

    class Opt:
        param_groups = [{"lr": 0.0}]
    update_learning_rate(Opt, 50000, 5e-4)
    assert Opt.param_groups[0]["lr"] == pytest.approx(2.5e-4)
    update_learning_rate(Opt, 0, 5e-4)
    assert Opt.param_groups[0]["lr"] == pytest.approx(5e-4)


def test_control_plane_is_not_restated():
    """The reference's epoch loop (train.py:38-169) and TrainParams (utils/train_utils.py:50-90) are out of scope
    (SURVEY 2 rows 9, 11): the package offers the per-batch pieces only and the reference's own loop drives them."""
    import dl_vqa_amd.train as T
    for name in ("train", "evaluate", "TrainParams", "get_train_params", "get_metrics", "get_zeroed_metrics_dict"):
        assert not hasattr(T, name), name
    for name in ("run_batch", "batch_accuracy", "update_learning_rate", "FusedAdam", "soft_ce_loss_and_score"):
        assert hasattr(T, name), name
