"""SURVEY.md 8f rank 2 (evaluation maths) and the matched losses of rank 4 against goldens produced by the reference's
own functions (tests/golden/eval_losses.npz, oracle/gen_golden.py:gen_eval_and_losses).  CPU: the torch-op forms;
GPU (-m gpu): the native launch (csrc/eval.hip) and the losses on device tensors.  Tolerance: 1e-4 relative
(north_star's floating-point bar), 1e-5 for the losses' f32 means."""
import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def G():
    return load_golden("eval_losses.npz")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _check_eval(G, dev):
    import fastposecnn_amd.lib  # noqa: F401
    import gpu_tensor_funcs as gtf
    q0, q1, sym = _t(G["q0"], dev), _t(G["q1"], dev), _t(G["sym"], dev)
    close = lambda got, key, rtol=1e-4, atol=1e-4: np.testing.assert_allclose(got.cpu().numpy(), G[key], rtol=rtol, atol=atol)
    mixed = gtf.get_quat_distance(q0, q1, sym)
    assert mixed.dtype == torch.float64
    close(mixed, "deg_mixed")                                   # non-symmetric pairs first, then symmetric (reference order)
    raw = gtf.get_quat_distance(q0, q1)
    assert raw.dtype == torch.float32
    close(raw, "deg_raw")
    assert abs(float(raw[3])) < 1e-3                             # antipodal pair
    close(gtf.get_symmetric_quat_distance(q0, q1), "deg_sym")
    close(gtf.get_quat_distance(q0, q1, torch.ones_like(sym)), "deg_all_sym")
    none_sym = gtf.get_quat_distance(q0, q1, torch.zeros_like(sym))
    assert none_sym.dtype == torch.float32
    close(none_sym, "deg_none_sym")
    assert gtf.get_quat_distance(q0[:0], q1[:0], sym[:0]).numel() == 0
    RT0, RT1, s0, s1 = _t(G["RT0"], dev), _t(G["RT1"], dev), _t(G["s0"], dev), _t(G["s1"], dev)
    close(gtf.get_3d_ious(RT0, RT1, s0, s1), "iou3d", rtol=2e-4, atol=1e-7)
    close(gtf.get_3d_ious(RT0, RT0, s0, s0), "iou3d_self", rtol=1e-4, atol=1e-7)
    close(gtf.from_Ts_get_offset_error(_t(G["T0"], dev), _t(G["T1"], dev)), "offset", rtol=1e-5, atol=1e-5)
    close(gtf.from_RTs_get_T_offset_errors(RT0, RT1), "offset_rt", rtol=1e-4, atol=1e-4)
    rot, ex = gtf.quat_symmetric_tf(q1[:2], q0[:2])
    assert tuple(rot.shape) == (2, 360, 4) and rot.dtype == torch.float64 and tuple(ex.shape) == (2, 360, 4)
    close(rot, "sym_tf_first2", rtol=1e-6, atol=1e-7)
    # APs
    cls = np.unique(G["cls"])
    raw_data = {k: {int(c): _t(G[f"raw_{k}_{c}"], dev) for c in cls} for k in ("degree_error", "3d_iou", "offset_error")}
    thr = {"degree_error": torch.tensor([5., 10., 30., 60.], device=dev), "3d_iou": torch.tensor([1., 10., 25., 50.], device=dev),
           "offset_error": torch.tensor([5., 10., 50., 200.], device=dev)}
    ops = {"degree_error": torch.less, "3d_iou": torch.greater, "offset_error": torch.less}
    aps = gtf.calculate_aps(raw_data, thr, ops)
    for k in aps:
        assert set(aps[k]) == set(int(c) for c in cls) | {"mean"}
        for c, v in aps[k].items():
            np.testing.assert_allclose(v.cpu().numpy(), G[f"aps_{k}_{c}"], rtol=1e-6)
    raw2 = {k: {c: torch.nan_to_num(v.double(), nan=1e9) for c, v in d.items()} for k, d in raw_data.items()}
    cthr = {"degree_error+offset_error": torch.vstack((torch.tensor([5, 10, 60]), torch.tensor([5, 50, 200]))).to(dev)}
    caps = gtf.calculate_complex_aps(raw2, cthr, ops)["degree_error+offset_error"]
    for c, v in caps.items():
        np.testing.assert_allclose(v.cpu().numpy(), G[f"caps_{c}"], rtol=1e-6)


def _check_losses(G, dev):
    import fastposecnn_amd.lib  # noqa: F401
    import loss as L
    n = G["q0"].shape[0]

    def matched():
        m = {"instance_masks": torch.zeros((2, n, 2, 2), device=dev), "symmetric_ids": _t(G["sym"], dev), "class_ids": _t(G["cls"], dev)}
        leaves = {}
        for key, a, b in (("quaternion", "q0", "q1"), ("xy", "xy0", "xy1"), ("z", "z0", "z1"), ("scales", "s0", "s1"),
                          ("R", "R0", "R1"), ("T", "T0", "T1"), ("RT", "RT0", "RT1")):
            p = _t(G[b], dev).clone().requires_grad_(True)
            leaves[key] = p
            m[key] = torch.stack((_t(G[a], dev), p))
        return m, leaves

    table = (("QLoss", L.QLoss(key="quaternion"), "quaternion"), ("XYLoss", L.XYLoss(key="xy"), "xy"), ("ZLoss", L.ZLoss(key="z"), "z"),
             ("ScalesLoss", L.ScalesLoss(key="scales"), "scales"), ("RLoss", L.RLoss(key="R"), "R"), ("TLoss", L.TLoss(key="T"), "T"),
             ("Iou3dLoss", L.Iou3dLoss(), "RT"), ("OffsetLoss", L.OffsetLoss(), "RT"))
    for name, fn, key in table:
        m, leaves = matched()
        val = fn(m)
        want = G[f"loss_{name}"]
        assert str(val.dtype).endswith(str(want.dtype)), (name, val.dtype, want.dtype)
        np.testing.assert_allclose(val.detach().cpu().numpy(), want, rtol=2e-5, atol=1e-6, err_msg=name)
        if f"grad_{name}" in G:
            val.backward()
            gw = G[f"grad_{name}"]
            np.testing.assert_allclose(leaves[key].grad.cpu().numpy(), gw, rtol=2e-3, atol=2e-5 * max(1e-6, np.abs(gw).max()), err_msg=name)
        assert bool(torch.isnan(fn(None)))
        assert bool(torch.isnan(fn({"instance_masks": m["instance_masks"]})))       # key absent
    ml = _t(G["mask_logits"], dev).requires_grad_(True)
    gt = _t(G["gt_mask"], dev)
    for name, fn in (("CE", L.CE()), ("CCE", L.CCE())):
        ml.grad = None
        val = fn({"logits": {"mask": ml}}, {"mask": gt})
        val.backward()
        np.testing.assert_allclose(val.detach().cpu().numpy(), G[f"loss_{name}"], rtol=1e-5)
        np.testing.assert_allclose(ml.grad.cpu().numpy(), G[f"grad_{name}"], rtol=1e-4, atol=1e-8)
    # Focal: pytorch_toolbelt's published formula on a hand-checkable case (parity unpinned: package absent)
    x = torch.tensor([[[[2.0]], [[-1.0]]]], device=dev)                       # one pixel, two classes, target class 0
    y = torch.zeros((1, 1, 1), dtype=torch.int64, device=dev)
    lp = torch.log_softmax(x, dim=1).double()
    want = 0.0
    for c, tgt in ((0, 1.0), (1, 0.0)):
        z = float(lp[0, c, 0, 0])
        bce = max(z, 0) - z * tgt + np.log1p(np.exp(-abs(z)))
        want += (1 - np.exp(-bce)) ** 2 * bce * (0.5 * tgt + 0.5 * (1 - tgt))
    got = L.Focal()({"logits": {"mask": x}}, {"mask": y})
    assert abs(float(got) - want) < 1e-6


def test_eval_maths_torch_forms_match_reference(G):
    _check_eval(G, "cpu")


def test_losses_match_reference_cpu(G):
    _check_losses(G, "cpu")


def test_total_loss_arithmetic():
    """F/lib/pose_regressor.py:265-307: NaN losses leave the weighted task sum; a task of only NaNs leaves the total."""
    import fastposecnn_amd.lib  # noqa: F401
    import loss as L
    crit = L.head_training_criterion()
    torch.manual_seed(0)
    ml = torch.randn(1, 7, 6, 8, requires_grad=True)
    out = {"logits": {"mask": ml}}
    batch = {"mask": torch.randint(0, 7, (1, 6, 8))}
    total, rep = L.total_loss(crit, out, batch, None)
    mask_sum = 5.0 * (rep["mask"]["loss_ce"] + rep["mask"]["loss_cce"] + rep["mask"]["loss_focal"])
    assert torch.allclose(total, mask_sum) and torch.allclose(rep["mask"]["task_total_loss"], mask_sum)
    for k in ("quaternion", "xy", "z", "scales"):
        assert bool(torch.isnan(rep[k]["task_total_loss"]))
    total.backward()
    assert ml.grad is not None and torch.isfinite(ml.grad).all()


@pytest.mark.gpu
def test_eval_maths_native_matches_reference(G):
    _check_eval(G, "cuda")


@pytest.mark.gpu
def test_losses_match_reference_gpu(G):
    _check_losses(G, "cuda")


def _check_metrics(G, dev):
    """lib/metrics.py accumulators over two rounds against the arithmetic of F/lib/metrics.py applied to the golden values."""
    import fastposecnn_amd.lib  # noqa: F401
    import metrics as M
    n = G["q0"].shape[0]
    m = {"symmetric_ids": _t(G["sym"], dev), "class_ids": _t(G["cls"], dev)}
    for key, a, b in (("quaternion", "q0", "q1"), ("scales", "s0", "s1"), ("T", "T0", "T1"), ("RT", "RT0", "RT1")):
        m[key] = torch.stack((_t(G[a], dev), _t(G[b], dev)))
    deg, iou, off, off_rt = G["deg_mixed"], G["iou3d"], G["offset"], float(G["offset_rt"])
    table = M.head_training_metrics()["pose"]
    assert set(table) == {"degree_error", "degree_error_AP_5", "iou_3d_mAP_0.25", "iou_3d_accuracy", "offset_error_AP_5cm", "offset_error"}
    for rounds in (1, 2):
        for e in table.values():
            e["F"].reset()
        out = {}
        for _ in range(rounds):
            out = {k: e["F"](m) for k, e in table.items()}
        run = lambda v: v / 2 if rounds == 1 else (v / 2 + v) / 2          # state = (state + value) / 2 from 0
        want = {"degree_error": run(deg.mean()), "degree_error_AP_5": (deg < 5).mean() * 100,
                "iou_3d_mAP_0.25": (iou > 0.25).mean() * 100, "iou_3d_accuracy": run((iou * 100).mean()),
                "offset_error_AP_5cm": (off < 5).mean() * 100, "offset_error": run(off_rt)}
        for k, w in want.items():
            assert abs(float(out[k]) - float(w)) <= 1e-4 * max(1.0, abs(float(w))), (k, float(out[k]), float(w))
    # no matches: the state is untouched
    d = M.DegreeError()
    assert float(d(None)) == 0.0 and float(d({"instance_masks": torch.zeros(1)})) == 0.0


def test_metrics_accumulators_cpu(G):
    _check_metrics(G, "cpu")


@pytest.mark.gpu
def test_metrics_accumulators_gpu(G):
    _check_metrics(G, "cuda")
