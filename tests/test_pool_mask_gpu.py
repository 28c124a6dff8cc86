"""MaxPoolMask / MaxPoolUndoMask (include/convnet_hip.h; csrc/pool_norm.hip: pool_fwd_max32_mask_kernel / pool_undo_max32_mask_kernel): the
3 x 3 stride-2 max pooling that records per pooled element which inputs of its window equal the maximum, and the undo that routes the
derivatives from those masks alone.  Checked (a) against the CPU oracle (the reference's MaxPool / MaxPoolUndo, src/CPUMatrix.cc:574-700)
and (b) bit for bit against the library's own MaxPool + MaxPoolUndo / MaxPoolUndoRelu pair, which it replaces on the fused host path —
on maps with odd and even sizes, with and without padding, with tie-heavy inputs (every tie must receive the derivative),
negative maxima (bit 9), accumulation into the target, and through MaxPoolEdge itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import rel_err  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from convnet_amd.matrix import Matrix
    from hip_adapter import HipImpl
    Matrix.SetupCUDADevice(0)
    return HipImpl()


def _run(g, x, dy, t0, st, relu):
    """(y, dx) by the mask pair and by the classic pair on the same device tensors"""
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat, _desc
    d = _desc(g, True)
    xm = _mat(x, g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))
    dym = _mat(dy, g.N, g.Mx * g.My * g.C, (g.N, g.Mx, g.My, g.C))
    out = []
    for masked in (True, False):
        ym = _mat(np.zeros(g.pooled_shape(), np.float32), g.N, g.Mx * g.My * g.C, (g.N, g.Mx, g.My, g.C))
        tm = _mat(t0.copy(), g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))
        if masked:
            mk = Matrix()
            mk.AllocateGPUMemory(g.N, (g.Mx * g.My * g.C + 1) // 2)
            assert Matrix.ConvMaxPoolMask(xm, ym, mk, d), "this geometry must have a mask kernel"
            Matrix.ConvMaxPoolUndoMask(dym, mk, tm, d, st, relu)
        else:
            Matrix.ConvMaxPool(xm, ym, d)
            (Matrix.ConvMaxPoolUndoRelu if relu else Matrix.ConvMaxPoolUndo)(xm, dym, ym, tm, d, st)
        out.append((ym.ToNumpy().reshape(g.pooled_shape()), tm.ToNumpy().reshape(g.in_shape())))
    return out


CASES = [
    Geom(N=32, C=8, H=21, W=21, F=8, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),     # pool1's form: odd map, padding 1 -> 11 x 11
    Geom(N=16, C=5, H=26, W=26, F=5, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),     # pool2's: even map -> 13 x 13
    Geom(N=8, C=3, H=11, W=11, F=3, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),      # pool5's -> 6 x 6
    Geom(N=4, C=2, H=9, W=13, F=2, Ky=3, Kx=3, sy=2, sx=2),                       # no padding, rectangular -> 4 x 6
    Geom(N=256, C=16, H=110, W=110, F=16, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),  # pool1's own map, 16 channels: the XCD block order at size
]
_id = lambda g: f"N{g.N}C{g.C}H{g.H}W{g.W}p{g.pady}"  # noqa: E731


@pytest.mark.parametrize("g", CASES, ids=_id)
@pytest.mark.parametrize("ties", [False, True])
def test_mask_pair_equals_classic_pair_and_oracle(hip, g, ties):
    rng = np.random.default_rng(61)
    if ties:   # small integers: most windows hold their maximum several times, many maxima are <= 0
        x = rng.integers(-2, 2, g.in_shape()).astype(np.float32)
    else:
        x = rng.standard_normal(g.in_shape()).astype(np.float32)
    dy = rng.standard_normal(g.pooled_shape()).astype(np.float32)
    t0 = rng.standard_normal(g.in_shape()).astype(np.float32)
    for st, relu in ((0.0, False), (0.0, True), (1.0, False)):
        (y_m, dx_m), (y_c, dx_c) = _run(g, x, dy, t0, st, relu)
        assert np.array_equal(y_m, y_c) and np.array_equal(dx_m, dx_c), (st, relu, float(np.abs(dx_m - dx_c).max()))
    # accumulating AND the fused ReLU': MaxPoolUndoRelu masks the accumulated target too — refused, nothing written
    with pytest.raises(Exception):
        _run(g, x, dy, t0, 1.0, True)
    if g.N * g.C * g.H * g.W <= 2 * 10 ** 6:
        y = oracle.port.max_pool(g, x)
        ref = oracle.port.max_pool_undo(g, x, dy, y)
        (y_m, dx_m), _ = _run(g, x, dy, t0, 0.0, False)
        assert np.array_equal(y_m, y) and rel_err(dx_m, ref) < 1e-6   # (a pixel that ties in several windows: the oracle adds them in its own order)


def test_geometries_without_a_mask_kernel_are_refused(hip):
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat, _desc
    for g in (Geom(N=8, C=2, H=8, W=8, F=2, Ky=2, Kx=2, sy=2, sx=2),                   # 2 x 2 windows
              Geom(N=6, C=2, H=9, W=9, F=2, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1)):  # N % 4 != 0
        xm = _mat(np.zeros(g.in_shape(), np.float32), g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))
        ym = _mat(np.full(g.pooled_shape(), 7.0, np.float32), g.N, g.Mx * g.My * g.C, (g.N, g.Mx, g.My, g.C))
        mk = Matrix()
        mk.AllocateGPUMemory(g.N, (g.Mx * g.My * g.C + 1) // 2)
        assert not Matrix.ConvMaxPoolMask(xm, ym, mk, _desc(g, True))
        assert np.all(ym.ToNumpy() == 7.0)   # refused = nothing written


def test_maxpool_edge_uses_the_mask_only_for_its_own_forward_pass(hip):
    """MaxPoolEdge with the fused entry points: ComputeDown after its own ComputeUp goes through the mask; handed OTHER matrices (a
    teacher-forced derivative check feeds its own states) it takes the reference's call — both give the classic result."""
    from convnet_amd import _lib, models, pbtxt
    from convnet_amd.convnet import ConvNet
    from convnet_amd.edge import MaxPoolEdge
    from convnet_amd.matrix import Matrix
    from hip_adapter import _mat
    net = ConvNet(pbtxt.parse(models.alexnet(image_size=64)), fused=True)
    e = next(x for x in net.edges_ if isinstance(x, MaxPoolEdge))
    assert e.fused
    g = Geom(N=8, C=e.num_input_channels_, H=e.image_size_y_, W=e.image_size_x_, F=e.num_input_channels_, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1)
    rng = np.random.default_rng(62)
    x = np.maximum(rng.standard_normal(g.in_shape()), 0).astype(np.float32)   # a ReLU layer's state
    dy = rng.standard_normal(g.pooled_shape()).astype(np.float32)
    xm = _mat(x, g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))
    ym = _mat(np.zeros(g.pooled_shape(), np.float32), g.N, g.Mx * g.My * g.C, (g.N, g.Mx, g.My, g.C))
    dym = _mat(dy, g.N, g.Mx * g.My * g.C, (g.N, g.Mx, g.My, g.C))
    dxm = _mat(np.zeros(g.in_shape(), np.float32), g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))
    y = oracle.port.max_pool(g, x)
    want = oracle.port.max_pool_undo(g, x, dy, y) * (x > 0)
    _lib.profile_enable(True)
    e.ComputeUp(xm, ym, True, train=True)
    e.ComputeDown(dym, xm, ym, dxm, True, fuse_mask=1.0)
    names = [r["kernel"] for r in _lib.profile_report()]
    _lib.profile_enable(False)
    assert "pool_fwd_mask_kernel<max>" in names and "pool_undo_mask_kernel<max>" in names, names
    assert np.array_equal(ym.ToNumpy().reshape(g.pooled_shape()), y) and rel_err(dxm.ToNumpy().reshape(g.in_shape()), want) < 1e-6
    masked = dxm.ToNumpy().copy()
    x2 = _mat(x, g.N, g.W * g.H * g.C, (g.N, g.W, g.H, g.C))   # the same values in another matrix: not what the mask was written for
    _lib.profile_enable(True)
    e.ComputeDown(dym, x2, ym, dxm, True, fuse_mask=1.0)
    names = [r["kernel"] for r in _lib.profile_report()]
    _lib.profile_enable(False)
    assert "pool_undo_mask_kernel<max>" not in names and any(n.startswith("pool_undo_kernel") for n in names), names
    assert np.array_equal(dxm.ToNumpy(), masked)   # the reference's call and the mask agree bit for bit


def test_alexnet_step_with_and_without_the_masks_is_bit_identical(hip):
    """The real AlexNet (224 x 224, 32 images, fused host path, dropout off so that both nets see the same units) trained for two steps
    twice from the same parameters and batch: with the mask pair on its three pooling edges, and with those edges on the reference's
    MaxPool / MaxPoolUndo(Relu) calls.  Every parameter, every gradient and every optimizer history must agree bit for bit — and so must the
    batched plain SGD step against one call per tensor (the second net also takes its optimizer steps one by one)."""
    from convnet_amd import _lib, models
    from convnet_amd.edge import MaxPoolEdge
    from test_net_gpu import build, copy_params
    text = models.alexnet(dropprob=0.0)
    a = build(text, 32, True, seed_data=9)
    b = build(text, 32, True, seed_data=9)
    copy_params(a, b)
    for e in b.edges_:
        if isinstance(e, MaxPoolEdge):
            e.fused = False            # the reference's call pair
    b_update = b.UpdateWeights

    def one_by_one():                  # ... and one optimizer call per tensor
        keep, b.fused = b.fused, False
        try:
            b_update()
        finally:
            b.fused = keep
    b.UpdateWeights = one_by_one
    _lib.profile_enable(True)
    for _ in range(2):
        a.TrainOneBatch()
    names_a = {r["kernel"] for r in _lib.profile_report()}
    _lib.profile_enable(False)
    _lib.profile_enable(True)
    for _ in range(2):
        b.TrainOneBatch()
    names_b = {r["kernel"] for r in _lib.profile_report()}
    _lib.profile_enable(False)
    assert "pool_undo_mask_kernel<max>" in names_a and "sgd_multi_kernel" in names_a, names_a
    assert "pool_undo_mask_kernel<max>" not in names_b and "sgd_multi_kernel" not in names_b and "sgd_kernel" in names_b, names_b
    assert np.array_equal(a.parameters_.ToNumpy(), b.parameters_.ToNumpy())
    for ea, eb in zip(a.edges_, b.edges_):
        if hasattr(ea, "grad_weights_"):
            assert np.array_equal(ea.GetGradWeight().ToNumpy(), eb.GetGradWeight().ToNumpy()), ea.GetName()
            assert np.array_equal(ea.weight_optimizer_.gradient_history_.ToNumpy(), eb.weight_optimizer_.gradient_history_.ToNumpy()), ea.GetName()
            if not ea.has_no_bias_:
                assert np.array_equal(ea.GetGradBias().ToNumpy(), eb.GetGradBias().ToNumpy()), ea.GetName()
