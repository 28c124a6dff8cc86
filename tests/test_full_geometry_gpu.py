"""GPU parity at the FULL geometry of BASELINE.json's configs — the launch-shape paths that only the real
sizes trigger (conv1 fprop: 12 100 pixels x 96-row tile; conv2 dgrad: 4 merged stride classes on the 55x55 input; conv2 fprop: tail split;
conv3 dgrad: split-K + reduce; the bias row of conv1/conv2 wgrad) are exercised by the real shapes, not by surrogates.

 * the real AlexNet model (models.alexnet() == examples/imagenet/CLS_net_20140621074703.pbtxt, pinned field for field on the
   CPU by tests/test_host_cpu.py), 224x224x3, a TRAINING pass (dropout on, device masks replayed on the CPU): every layer's
   activation, every layer's derivative, every edge's dW/db against the CPU oracle (oracle.port, pinned to the reference
   build; once against oracle.ref = the reference's own compiled code), fused and unfused, N = 4, 8 and the benchmark's 256;
 * every AlexNet conv layer at N = 256: the three kernels are mutually adjoint (<conv(x,w),dy> = <x,convT(dy,w)> = <w,wgrad(x,dy)>,
   float64 on the host) and sampled outputs of each equal a float64 evaluation of the reference's definition
   (cudamat_conv_gemm.cuh:5-10 layout, src/edge.cc:108-114 sizes); fused bias+ReLU == the unfused sequence bit for bit;
 * configs[0] / configs[1]: mnist-conv at its batch 100 and the LeNet-5-class net at batch 128, whole net vs oracle;
 * the reference's own CPU HOST (ConvNet::Fprop/ComputeDeriv/Bprop over CPUMatrix) on the full AlexNet at N = 4:
   committed samples of its gradient (tests/golden/ref_host_alexnet224.npz, tests/golden/make_ref_host_alexnet_golden.py).

Tolerance: the reference's own criterion max|a-b| / mean|a+b| < 1e-4 (py/test_conv.py:382-392)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Geom  # noqa: E402
from golden_cases import rel_err  # noqa: E402
from fp64_ref import ref_up as _ref_up, ref_down as _ref_down, ref_outp as _ref_outp  # noqa: E402

TOL = 1e-4
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    from convnet_amd.matrix import Matrix
    Matrix.SetupCUDADevice(0)
    return Matrix


@pytest.fixture(scope="module")
def hip(gpu):
    from hip_adapter import HipImpl
    return HipImpl()


def _train_pass(net):
    for l in net.layers_:
        l.ResetAddOrOverwrite()
    net.GetBatch(net.train_dataset_)
    x = net.input_layers_[0].GetState().ToNumpy()
    labels = net.output_layers_[0].GetData().ToNumpy().reshape(-1)
    net.Fprop(True)
    net.ComputeDeriv()
    net.Bprop()
    states = {l.GetName(): l.GetState().ToNumpy().reshape(-1) for l in net.layers_}
    derivs = {l.GetName(): l.GetDeriv().ToNumpy().reshape(-1) for l in net.layers_ if not l.IsInput()}
    return x, labels, states, derivs


def _within(got, want, tol, ulps=4):
    """The reference's metric (max|a-b| / mean|a+b| < tol) — or, for the elements that miss it, agreement to `ulps` fp32 ulps of
    the element itself.  The metric divides the LARGEST error by the MEAN magnitude: on a heavy-tailed tensor (fc8's dW at N = 4:
    a four-term sum per element, the label column ~3000x the mean) one ulp of the largest element is already 1.8e-4 of the mean,
    so only an implementation that rounds bit-identically to the CPU loop can pass it there; any other summation order or product
    formation (the bf16-split path, an FMA) differs in the last bit of some large element."""
    a, b = np.asarray(got, np.float64).ravel(), np.asarray(want, np.float64).ravel()
    d = np.abs(a - b)
    miss = d >= tol * (np.abs(a + b).mean() + 1e-30)
    return bool(np.all(d[miss] <= ulps * 2.0 ** -23 * np.maximum(np.abs(a[miss]), np.abs(b[miss]))))


def _check_whole_net(net, impl, forced_only=False, chain_tol=3e-4):
    """Three comparisons of one training pass with the CPU oracle:
      1. forward chain, un-forced: the oracle sees only the input, the parameters and the device's dropout masks; every layer's
         state within TOL (the softmax output element-wise: its large probabilities dominate a max-over-mean metric);
      2. backward, teacher-forced op by op: every backward op is fed the device's own states and incoming derivative and must
         reproduce the device's output within TOL;
      3. backward, end to end (unless ``forced_only``): the oracle propagates ITS OWN derivatives from the loss to conv1 — only the
         gates (which units a ReLU passed, where a max-pool found its maximum) are read from the device's states.  With ~1.5 M ReLU
         units per image a handful sit within fp32 rounding of zero and gate differently on two fp32 machines; one such flip in
         fc7 re-colours that image's whole upstream derivative by ~1 %, so an end-to-end comparison that lets each side gate for
         itself measures the flips, not the kernels (the reference's own GPU and CPU builds differ the same way).  Accumulated
         over the 13 backward ops the bound is CHAIN_TOL."""
    from oracle_net import forward_backward
    CHAIN_TOL = chain_tol
    x, labels, states, derivs = _train_pass(net)
    t0 = time.time()
    acts, od, og = forward_backward(net, x, labels, impl=impl, force=(states, derivs), dropout_states=states)
    for l in net.layers_:
        got, want = states[l.GetName()], acts[l.GetName()]
        if l.IsOutput():
            assert np.allclose(got, want, rtol=5e-4, atol=1e-10), ("output probabilities", float(np.abs(got / np.maximum(want, 1e-30) - 1).max()))
        else:
            e = rel_err(got, want)
            assert e < TOL, ("state", l.GetName(), e)
    for name, d in od.items():
        if name in derivs and not net.GetLayerByName(name).IsInput():
            e = rel_err(derivs[name], d)
            assert e < TOL, ("deriv (forced)", name, e)
    grads = {}
    for e in net.edges_:
        if e.GetName() in og:
            dw, db = og[e.GetName()]
            grads[e.GetName()] = (e.GetGradWeight().ToNumpy().reshape(-1), e.GetGradBias().ToNumpy().reshape(-1))
            # the ulp clause of _within is for FC edges only (fc8's label column at small N); conv edges meet the plain metric
            ok = _within(grads[e.GetName()][0], dw, TOL) if type(e).__name__ == "FCEdge" else rel_err(grads[e.GetName()][0], dw) < TOL
            assert ok, ("dW (forced)", e.GetName(), rel_err(grads[e.GetName()][0], dw))
            assert rel_err(grads[e.GetName()][1], db) < TOL, ("db (forced)", e.GetName(), rel_err(grads[e.GetName()][1], db))
    if not forced_only:
        _, ud, ug = forward_backward(net, x, labels, impl=impl, force=(states, None), dropout_states=states)
        worst = 0.0
        for name, d in ud.items():
            if name in derivs and not net.GetLayerByName(name).IsInput():
                e = rel_err(derivs[name], d)
                worst = max(worst, e)
                assert e < CHAIN_TOL, ("deriv (end to end, device gates)", name, e)
        for name, (dw, db) in ug.items():
            e = max(rel_err(grads[name][0], dw), rel_err(grads[name][1], db))
            worst = max(worst, e)
            assert e < CHAIN_TOL, ("dW/db (end to end, device gates)", name, e)
        print(f"end-to-end backward (device gates): worst rel err {worst:.2e}")
    return time.time() - t0


def _build(text, batch, fused, seed=5):
    from test_net_gpu import build
    return build(text, batch, fused, seed_data=seed)


@pytest.mark.parametrize("N,fused,which,path", [(4, False, "port", "split"), (4, True, "port", "split"), (8, True, "port", "split"),
                                                (4, False, "ref", "split"), (4, False, "port", "fp32"), (8, True, "ref", "fp32")])
def test_real_alexnet_224_training_pass_vs_cpu_oracle(gpu, N, fused, which, path):
    """`path`: both ways the library forms its GEMM products (convnet_hip_set_matrix_path) — bf16-split (default) and fp32 MFMA."""
    from convnet_amd import _lib, models
    impl = oracle.port if which == "port" else oracle.ref
    if impl is None:
        pytest.skip("oracle/_ref not built on this box")
    _lib.lib.convnet_hip_set_matrix_path(1 if path == "split" else 0)
    try:
        net = _build(models.alexnet(), N, fused)
        assert net.input_layers_[0].GetSizeY() == 224 and net.GetLayerByName("hidden6").dropprob_ > 0
        _check_whole_net(net, impl)
    finally:
        _lib.lib.convnet_hip_set_matrix_path(1)


@pytest.mark.parametrize("N,path", [(32, "split"), (64, "split"), (32, "fp32")])
def test_real_alexnet_224_at_per_gpu_batches_of_strong_scaling(gpu, N, path):
    """The whole training pass at 32 / 64 images per GPU (global 256 over 8 / 4 GPUs): multi-pixel wave-columns in every conv layer."""
    from convnet_amd import _lib, models
    _lib.lib.convnet_hip_set_matrix_path(1 if path == "split" else 0)
    try:
        net = _build(models.alexnet(), N, True)
        _check_whole_net(net, oracle.port, forced_only=True)
    finally:
        _lib.lib.convnet_hip_set_matrix_path(1)


def test_real_alexnet_224_at_the_benchmark_batch_256(gpu):
    """BASELINE configs[2] as benchmarked: bs = 256 (two 128-image wave-columns per pixel, every split / tail-split / merged-class
    launch of bench.py), fused entry points, dropout on.  Backward teacher-forced (the un-forced variant runs at N = 4 and 8)."""
    from convnet_amd import models
    net = _build(models.alexnet(), 256, True)
    took = _check_whole_net(net, oracle.port, forced_only=True)
    print(f"oracle forward+backward at N=256: {took:.1f} s")


@pytest.mark.parametrize("which,N,fused", [("mnist_conv", 100, False), ("mnist_conv", 100, True), ("lenet5", 128, False), ("lenet5", 128, True)])
def test_config0_and_config1_nets_at_their_batch_sizes(gpu, which, N, fused):
    """examples/mnist-conv (bs 100, net.pbtxt / train.pbtxt) and the LeNet-5-class 28x28 net at bs 128 (BASELINE configs 0-1)."""
    from convnet_amd import models
    net = _build({"mnist_conv": models.mnist_conv, "lenet5": models.lenet5}[which](), N, fused)
    _check_whole_net(net, oracle.port)


# ---- per-layer, exact AlexNet sizes, N = 256 -----------------------------------------------------------------------------
def _alex_geoms(N=256):
    """The conv geometries of the real model at N images, read off the built graph."""
    from convnet_amd import models, pbtxt
    from convnet_amd.edge import ConvEdge
    from convnet_amd.convnet import ConvNet
    net = ConvNet(pbtxt.parse(models.alexnet()))
    out = {}
    for e in net.edges_:
        if isinstance(e, ConvEdge):
            s, d = e.GetSource(), e.conv_desc_
            out[e.GetDest().GetName()] = Geom(N, s.GetNumChannels(), s.GetSizeY(), s.GetSizeX(), d.num_output_channels, d.kernel_size_y,
                                              d.kernel_size_x, d.stride_y, d.stride_x, -d.padding_y, -d.padding_x)
    return out


def _dot64(a, b, chunk=1 << 24):
    a, b = a.reshape(-1), b.reshape(-1)
    return float(sum(np.dot(a[i:i + chunk].astype(np.float64), b[i:i + chunk].astype(np.float64)) for i in range(0, a.size, chunk)))


@pytest.mark.parametrize("layer", ["conv1", "conv2", "conv3", "conv4", "conv5"])
def test_alexnet_conv_layer_at_full_size_n256(hip, layer):
    _check_conv_layer_at_full_size(hip, layer, 256)


@pytest.mark.parametrize("layer,N", [("conv1", 32), ("conv2", 32), ("conv3", 32), ("conv4", 32), ("conv5", 32),
                                     ("conv1", 64), ("conv3", 64), ("conv5", 64), ("conv2", 128), ("conv4", 128),
                                     ("conv2", 96), ("conv3", 20), ("conv5", 20)])
def test_alexnet_conv_layer_at_full_size_per_gpu_batches_of_strong_scaling(hip, layer, N):
    """SURVEY 8(d) config 4 (src/convnet.cc:429-431): a global batch of 256 over 2 / 4 / 8 GPUs is 128 / 64 / 32 images per GPU.
    The GEMM column space is flat (GGParams::NP): below 128 images a wave-column spans several output pixels, a block tile up to
    eight (64 at N = 4), and border-tap skipping works on the union rectangle of the tile's pixels.  96 and 20 do not divide the
    wave-column: pixels straddle wave-columns and tiles."""
    _check_conv_layer_at_full_size(hip, layer, N)


def _check_conv_layer_at_full_size(hip, layer, N):
    g = _alex_geoms(N)[f"hidden{layer[-1]}_conv"]
    # conv1 7x7 s2 p1 -> 110x110; conv2 5x5 s2 -> 26x26; conv3/4 3x3 p1 13x13; conv5 3x3 p0 -> 11x11 (src/edge.cc:108-114)
    assert (g.My, g.Mx) == {"conv1": (110, 110), "conv2": (26, 26), "conv3": (13, 13), "conv4": (13, 13), "conv5": (11, 11)}[layer]
    rng = np.random.default_rng(40 + int(layer[-1]))
    x = rng.standard_normal(g.in_shape(), dtype=np.float32)
    w = rng.standard_normal(g.filt_shape(), dtype=np.float32) * np.float32(0.05)
    dy = rng.standard_normal(g.out_shape(), dtype=np.float32)
    y = hip.conv_up(g, x, w)
    dx = hip.conv_down(g, dy, w)
    dw = hip.conv_outp(g, x, dy)
    a, b, c = _dot64(y, dy), _dot64(x, dx), _dot64(w, dw)
    # (44 M to 2.4 G random-sign products per inner product: the three fp32 results agree to a few 1e-6 of |a| after cancellation;
    # a wrong tap, border or class shifts them by 1e-3 or more)
    # (denominator: |a|, or its typical size |y||dy|/sqrt(D) when this seed's inner product happens to cancel to something small)
    den = max(abs(a), (_dot64(y, y) * _dot64(dy, dy) / y.size) ** 0.5)
    assert abs(a - b) / den < 3e-5 and abs(a - c) / den < 3e-5, (a, b, c)
    scale_y, scale_dx, scale_dw = float(np.abs(y).mean()), float(np.abs(dx).mean()), float(np.abs(dw).mean())
    for _ in range(48):
        n, f, oy, ox = rng.integers(g.N), rng.integers(g.F), rng.integers(g.My), rng.integers(g.Mx)
        assert abs(_ref_up(g, x, w, f, oy, ox, n) - y[f, oy, ox, n]) < TOL * scale_y, ("fprop", f, oy, ox, n)
        cc, iy, ix = rng.integers(g.C), rng.integers(g.H), rng.integers(g.W)
        assert abs(_ref_down(g, dy, w, cc, iy, ix, n) - dx[cc, iy, ix, n]) < TOL * scale_dx, ("dgrad", cc, iy, ix, n)
    # corners and borders explicitly (padding taps, first/last stride class)
    for (oy, ox) in ((0, 0), (g.My - 1, g.Mx - 1), (0, g.Mx - 1)):
        assert abs(_ref_up(g, x, w, 1, oy, ox, g.N - 1) - y[1, oy, ox, g.N - 1]) < TOL * scale_y
    for (iy, ix) in ((0, 0), (g.H - 1, g.W - 1), (g.H - 1, 0), (1, g.W - 2)):
        assert abs(_ref_down(g, dy, w, g.C - 1, iy, ix, 0) - dx[g.C - 1, iy, ix, 0]) < TOL * scale_dx, ("dgrad border", iy, ix)
    for _ in range(6):
        cc, ky, kx, f = rng.integers(g.C), rng.integers(g.Ky), rng.integers(g.Kx), rng.integers(g.F)
        assert abs(_ref_outp(g, x, dy, cc, ky, kx, f) - dw[cc, ky, kx, f]) < TOL * scale_dw, ("wgrad", cc, ky, kx, f)
    # fused bias + ReLU epilogue == conv, then add_row_vec, then lower_bound (conv_edge.cc:145-148, layer.cc:549-551), bit for bit
    bias = rng.standard_normal(g.F, dtype=np.float32)
    fused = hip.conv_up_bias_relu(g, x, w, bias, relu=True)
    assert np.array_equal(fused, np.maximum(y + bias[:, None, None, None], np.float32(0)))


def test_reference_cpu_host_gradient_on_the_full_alexnet_golden(gpu):
    """tests/golden/ref_host_alexnet224.npz holds what the REFERENCE'S OWN HOST (src/convnet.cc ... over CPUMatrix/eigenmat, compiled
    unmodified) computed for the full 224x224 AlexNet at N = 4 (dropout off: the two RNGs cannot be matched) from hash-generated
    parameters and batch: per-edge gradient samples at fixed indices and per-edge gradient norms.  The
    product host must reproduce them, fused and unfused."""
    from convnet_amd import models
    from convnet_amd.convnet import ConvNet
    from test_reference_host import HashDataHandler
    import ref_host
    path = os.path.join(HERE, "golden", "ref_host_alexnet224.npz")
    G = np.load(path)
    N, seed = int(G["cfg"][0]), int(G["cfg"][1])
    text = models.alexnet(dropprob=0.0)
    for fused in (False, True):
        net = ConvNet(text, fused=fused)
        net.SetBatchsize(N)
        net.SetupDataset(HashDataHandler(net, N, 1, seed))
        net.AllocateMemory(False)
        total = net.parameters_.GetNumEls()
        assert total == int(G["total"])
        slices = [(off, n, e.GetDest().GetNumChannels()) for e, (off, n) in net.edge_slices_.items()]
        net.parameters_.FromNumpy(ref_host.golden_params(total, seed, slices))
        for l in net.layers_:
            l.ResetAddOrOverwrite()
        net.GetBatch(net.train_dataset_)
        net.Fprop(True)
        net.ComputeDeriv()
        net.Bprop()
        g = net.grad_parameters_.ToNumpy().reshape(-1)
        if not fused:   # forward-only scalar: the CE loss the reference host read at these parameters (its Layer::GetLoss)
            loss = sum(l.GetLoss() for l in net.output_layers_)
            assert abs(loss - float(G["loss"])) < 1e-4 * abs(float(G["loss"])), ("loss", loss, float(G["loss"]))
        for e in net.edges_:
            if e not in net.edge_slices_:
                continue
            off, n = net.edge_slices_[e]
            name = e.GetName().replace(":", "__")
            idx = G[f"idx_{name}"]
            got, want = g[off + idx].astype(np.float64), G[f"g_{name}"].astype(np.float64)
            # Each side gates its ~6 M ReLU units / pool windows for itself here (nothing of the reference's run but its
            # gradient is on file), so a few units within fp32 rounding of zero gate differently and every such flip shifts a
            # whole image's upstream derivative by ~1 % (see _check_whole_net): agreement is statistical — relative L2 error
            # of the samples — and 100x tighter than any real defect (a wrong tap, scale or border) would leave it.
            l2 = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            assert l2 < 1e-2, ("gradient samples", e.GetName(), fused, l2)
            nrm = float(np.linalg.norm(g[off:off + n].astype(np.float64)))
            assert abs(nrm - float(G[f"norm_{name}"])) < 2e-3 * float(G[f"norm_{name}"]), ("gradient norm", e.GetName(), fused)


# ---- BASELINE configs[4]: the VGG-style 3x3-stacked net, 224x224 ---------------------------------------
@pytest.mark.parametrize("fused,path", [(True, "split"), (False, "fp32")])
def test_vgg_224_training_pass_vs_cpu_oracle(gpu, fused, path):
    """models.vgg() (13 conv 3x3-s1-p1 layers on 224..14-pixel maps, five 2x2-s2 max-pools, 3 FC, dropout on) at 224 x 224, N = 4:
    every state, every derivative (teacher-forced and end to end with device gates), every dW / db against the CPU oracle —
    the 224^2 x 64 layers, the 2x2 pool kernels and the many-tile launches only these shapes produce."""
    from convnet_amd import _lib, models
    _lib.lib.convnet_hip_set_matrix_path(1 if path == "split" else 0)
    try:
        net = _build(models.vgg(), 4, fused)
        assert net.input_layers_[0].GetSizeY() == 224 and len([e for e in net.edges_ if type(e).__name__ == "ConvEdge"]) == 13
        # end-to-end bound: 21 backward ops deep instead of AlexNet's 13, the early ones sum 50 176 pixels per weight (measured
        # 4.1e-4 .. 6.5e-4 at conv3_x over runs and paths; the teacher-forced comparison above holds every op to 1e-4)
        took = _check_whole_net(net, oracle.port, chain_tol=1e-3)
        print(f"oracle forward+backward, VGG N=4: {took:.1f} s")
    finally:
        _lib.lib.convnet_hip_set_matrix_path(1)


@pytest.mark.parametrize("layer", ["conv1_1", "conv1_2"])
def test_vgg_first_conv_layers_at_full_size_n128(hip, layer):
    """conv1_1 (3 -> 64) and conv1_2 (64 -> 64), 3x3 s1 p1 on the 224 x 224 map at the benchmark's N = 128 (BASELINE configs[4]):
    50 176 output pixels x 128 images = 25 088 block tiles per launch; adjointness of the three kernels in float64 and float64
    spot checks incl. every corner, as for the AlexNet layers."""
    C = {"conv1_1": 3, "conv1_2": 64}[layer]
    g = Geom(128, C, 224, 224, 64, 3, 3, 1, 1, 1, 1)
    rng = np.random.default_rng(C)
    x = rng.standard_normal(g.in_shape(), dtype=np.float32)
    w = rng.standard_normal(g.filt_shape(), dtype=np.float32) * np.float32(0.05)
    dy = rng.standard_normal(g.out_shape(), dtype=np.float32)
    y = hip.conv_up(g, x, w)
    dx = hip.conv_down(g, dy, w)
    dw = hip.conv_outp(g, x, dy)
    a, b, c = _dot64(y, dy), _dot64(x, dx), _dot64(w, dw)
    den = max(abs(a), (_dot64(y, y) * _dot64(dy, dy) / y.size) ** 0.5)
    # (every dW element is a 6.4 M-term fp32 reduction — 50 176 pixels x 128 images — so <w, dW> carries ~1e-4 of accumulation
    # rounding where the AlexNet layers carry 1e-5; the float64 spot checks below pin individual elements)
    assert abs(a - b) / den < 3e-5 and abs(a - c) / den < 1.5e-4, (a, b, c)
    sy_, sdx, sdw = float(np.abs(y).mean()), float(np.abs(dx).mean()), float(np.abs(dw).mean())
    pts = [(0, 0), (223, 223), (0, 223), (223, 0), (1, 222)] + [(int(rng.integers(224)), int(rng.integers(224))) for _ in range(24)]
    for (py, px) in pts:
        n, f, cc = int(rng.integers(g.N)), int(rng.integers(g.F)), int(rng.integers(g.C))
        assert abs(_ref_up(g, x, w, f, py, px, n) - y[f, py, px, n]) < TOL * sy_, ("fprop", f, py, px, n)
        assert abs(_ref_down(g, dy, w, cc, py, px, n) - dx[cc, py, px, n]) < TOL * sdx, ("dgrad", cc, py, px, n)
    for _ in range(4):
        cc, ky, kx, f = int(rng.integers(g.C)), int(rng.integers(3)), int(rng.integers(3)), int(rng.integers(g.F))
        assert abs(_ref_outp(g, x, dy, cc, ky, kx, f) - dw[cc, ky, kx, f]) < TOL * sdw, ("wgrad", cc, ky, kx, f)


def test_vgg_pool1_2x2_stride2_at_full_size_n128(hip):
    """pool1 of the VGG net at N = 128: 2x2 windows, stride 2, 64 channels, 224 -> 112; forward bit-exact against numpy, undo routes
    each gradient to its window's maxima (ties: to every tied element, CPUMatrix.cc / cudamat_conv.cu semantics pinned in
    test_hip_parity.py) — checked on tie-free random data against the one-hot scatter."""
    g = Geom(128, 64, 224, 224, 64, 2, 2, 2, 2, 0, 0)
    rng = np.random.default_rng(12)
    x = rng.standard_normal(g.in_shape(), dtype=np.float32)
    y = hip.max_pool(g, x)
    win = x.reshape(64, 112, 2, 112, 2, 128)
    want = win.max(axis=(2, 4))
    assert y.shape == want.shape and np.array_equal(y, want)
    dy = rng.standard_normal(want.shape, dtype=np.float32)
    dx = hip.max_pool_undo(g, x, dy, y)
    hit = win == want[:, :, None, :, None, :]   # (a handful of the 103 M windows tie in float32: every tied element gets the gradient)
    assert want.size <= hit.sum() < want.size + 100
    ref = (hit * dy[:, :, None, :, None, :]).reshape(x.shape).astype(np.float32)
    assert np.array_equal(dx, ref)
