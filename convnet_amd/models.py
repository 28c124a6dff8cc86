"""Model definitions in the reference's pbtxt schema (proto/convnet_config.proto), generated as
text so they can also be written to disk and fed to the reference's own binaries.

* ``alexnet()``    — the AlexNet-class ILSVRC model of examples/imagenet/CLS_net_20140621074703.pbtxt
                     (BASELINE configs[2]/[3]): 224x224x3 -> conv7s2p1/96 -> max3s2p1 -> rnorm ->
                     conv5s2/256 -> max -> rnorm -> conv3p1/384 -> conv3p1/384 -> conv3/256 -> max ->
                     fc4096(drop .4) -> fc4096(drop .4) -> softmax1000; 62,357,608 parameters.
* ``mnist_conv()`` — examples/mnist-conv/net.pbtxt (configs[0]/[1]).
* ``vgg()``        — a VGG-style stack of 3x3 s1 p1 convs (configs[4]; no such pbtxt exists in the
                     reference).
tests/test_models.py checks the first two against the reference's files when they are mounted.
"""

_OPT_W = """  weight_optimizer {{
    epsilon: {eps}
    initial_momentum : 0.5
    final_momentum : {mom}
    momentum_transition_timescale : {tau}
    l2_decay: {l2}{extra}
  }}
  bias_optimizer {{
    epsilon: {eps}
    initial_momentum : 0.5
    final_momentum : {mom}
    momentum_transition_timescale : {tau}
  }}
"""


def _layer(name, channels, activation=None, dropprob=0.0, size=None, extra=""):
    s = f'layer {{\n  name: "{name}"\n  num_channels: {channels}\n'
    if size:
        s += f"  image_size_y: {size}\n  image_size_x: {size}\n"
    if activation:
        s += f"  activation: {activation}\n"
    if dropprob:
        s += f"  dropprob: {dropprob}\n"
    return s + extra + "}\n\n"


def _conv(src, dst, k, stride=1, pad=0, eps=0.01, mom=0.9, tau=2000, l2=0.0005, init_wt=1.0, init_bias=0.0, grad_check=""):
    return (f'edge {{\n  source: "{src}"\n  dest: "{dst}"\n  edge_type: CONVOLUTIONAL\n  kernel_size: {k}\n  stride : {stride}\n'
            f"  padding: {pad}\n  shared_bias: true\n  initialization: DENSE_UNIFORM_SQRT_FAN_IN\n  init_wt: {init_wt}\n"
            f"  init_bias: {init_bias}\n" + _OPT_W.format(eps=eps, mom=mom, tau=tau, l2=l2, extra="") + grad_check + "}\n\n")


def _fc(src, dst, eps=0.01, mom=0.9, tau=2000, l2=0.0005, init_wt=1.0, init_bias=0.0, norm_limit=0.0, grad_check=""):
    extra = f"\n    weight_norm_limit: {norm_limit}" if norm_limit else ""
    return (f'edge {{\n  source: "{src}"\n  dest: "{dst}"\n  edge_type: FC\n  initialization: DENSE_UNIFORM_SQRT_FAN_IN\n'
            f"  init_wt: {init_wt}\n  init_bias: {init_bias}\n" + _OPT_W.format(eps=eps, mom=mom, tau=tau, l2=l2, extra=extra) + grad_check + "}\n\n")


def _nin(src, dst, eps=0.01, mom=0.9, tau=2000, init_wt=1.0, grad_check=""):
    """CONV_ONETOONE (1x1 conv) with the unit-norm row constraint the reference's NIN model uses."""
    return (f'edge {{\n  source: "{src}"\n  dest: "{dst}"\n  edge_type: CONV_ONETOONE\n  initialization: DENSE_UNIFORM_SQRT_FAN_IN\n'
            f"  init_wt: {init_wt}\n" + _OPT_W.format(eps=eps, mom=mom, tau=tau, l2=0.0, extra="\n    weight_norm_constraint: 1") + grad_check + "}\n\n")


def _pool(src, dst, k, stride, pad=0, kind="MAXPOOL"):
    return (f'edge {{\n  source: "{src}"\n  dest: "{dst}"\n  edge_type: {kind}\n  kernel_size: {k}\n  stride : {stride}\n'
            f"  padding: {pad}\n}}\n\n")


def _rnorm(src, dst, add_scale=0.0005, pow_scale=0.75, frac=0.25):
    return (f'edge {{\n  source: "{src}"\n  dest: "{dst}"\n  edge_type: RESPONSE_NORM\n  add_scale: {add_scale}\n'
            f"  pow_scale: {pow_scale}\n  frac_of_filters_response_norm: {frac}\n}}\n\n")


def _header(name, seed=42):
    return f'name: "{name}"\nseed: {seed}\nmax_iter: 10000000\nprint_after: 100\n\n'


def _gc(grad_check, num_params=10):
    if not grad_check:
        return ""
    # several step sizes, largest first: fp32 loss round-off (~ulp(L)/(2 eps batch)) favours large steps, ReLU /
    # max-pool kinks favour small ones; the checker passes an edge if ANY epsilon passes (src/grad_check.cc:60-64)
    return (f"  grad_check: true\n  grad_check_num_params: {num_params}\n  grad_check_epsilon: 0.03\n  grad_check_epsilon: 0.01\n"
            "  grad_check_epsilon: 0.003\n  grad_check_epsilon: 0.001\n")


def alexnet(image_size=224, num_classes=1000, dropprob=0.4, grad_check=False):
    gc = _gc(grad_check)
    s = _header("CLS_net")
    s += _layer("input", 3, size=image_size)
    s += _layer("hidden1_conv", 96, "RECTIFIED_LINEAR") + _layer("hidden1_maxpool", 96) + _layer("hidden1_rnorm", 96, "RECTIFIED_LINEAR")
    s += _layer("hidden2_conv", 256, "RECTIFIED_LINEAR") + _layer("hidden2_maxpool", 256) + _layer("hidden2_rnorm", 256, "RECTIFIED_LINEAR")
    s += _layer("hidden3_conv", 384, "RECTIFIED_LINEAR") + _layer("hidden4_conv", 384, "RECTIFIED_LINEAR")
    s += _layer("hidden5_conv", 256, "RECTIFIED_LINEAR") + _layer("hidden5_maxpool", 256)
    s += _layer("hidden6", 4096, "RECTIFIED_LINEAR", dropprob) + _layer("hidden7", 4096, "RECTIFIED_LINEAR", dropprob)
    s += _layer("output", num_classes, "SOFTMAX")
    s += _conv("input", "hidden1_conv", 7, 2, 1, grad_check=gc)
    s += _pool("hidden1_conv", "hidden1_maxpool", 3, 2, 1) + _rnorm("hidden1_maxpool", "hidden1_rnorm")
    s += _conv("hidden1_rnorm", "hidden2_conv", 5, 2, 0, init_bias=1.0, grad_check=gc)
    s += _pool("hidden2_conv", "hidden2_maxpool", 3, 2, 1) + _rnorm("hidden2_maxpool", "hidden2_rnorm")
    s += _conv("hidden2_rnorm", "hidden3_conv", 3, 1, 1, grad_check=gc)
    s += _conv("hidden3_conv", "hidden4_conv", 3, 1, 1, init_bias=1.0, grad_check=gc)
    s += _conv("hidden4_conv", "hidden5_conv", 3, 1, 0, init_bias=1.0, grad_check=gc)
    s += _pool("hidden5_conv", "hidden5_maxpool", 3, 2, 1)
    s += _fc("hidden5_maxpool", "hidden6", norm_limit=4, grad_check=gc) + _fc("hidden6", "hidden7", norm_limit=4, grad_check=gc)
    s += _fc("hidden7", "output", norm_limit=4, grad_check=gc)
    return s


def alexnet_nin(image_size=224, num_classes=1000, grad_check=False, dropout=True):
    """examples/imagenet/CLS_net_20140801232522.pbtxt: the network-in-network variant — 1x1
    CONV_ONETOONE layers after conv2..conv5, 512-channel conv5, dropout 0.1/0.3/0.5."""
    gc = _gc(grad_check)
    R = "RECTIFIED_LINEAR"
    d = (lambda p: p) if dropout else (lambda p: 0.0)
    s = _header("CLS_net", seed=80638)
    s += _layer("input", 3, size=image_size)
    s += _layer("hidden1_conv", 96, R) + _layer("hidden1_maxpool", 96) + _layer("hidden1_rnorm", 96, R)
    s += _layer("hidden2_conv", 256, R) + _layer("hidden2_conv_nin1", 256, R) + _layer("hidden2_maxpool", 256) + _layer("hidden2_rnorm", 256, R)
    s += _layer("hidden3_conv", 384, R) + _layer("hidden3_conv_nin1", 768, R)
    s += _layer("hidden4_conv", 384, R) + _layer("hidden4_conv_nin1", 768, R, d(0.1)) + _layer("hidden4_conv_nin2", 384, R)
    s += _layer("hidden5_conv", 512, R) + _layer("hidden5_conv_nin1", 1024, R, d(0.3)) + _layer("hidden5_conv_nin2", 512, R)
    s += _layer("hidden5_maxpool", 512)
    s += _layer("hidden6", 4096, R, d(0.5)) + _layer("hidden7", 4096, R, d(0.5)) + _layer("output", num_classes, "SOFTMAX")
    s += _conv("input", "hidden1_conv", 7, 2, 1, l2=0.0, grad_check=gc)
    s += _pool("hidden1_conv", "hidden1_maxpool", 3, 2, 1) + _rnorm("hidden1_maxpool", "hidden1_rnorm")
    s += _conv("hidden1_rnorm", "hidden2_conv", 5, 2, 1, l2=0.0, init_bias=1.0, grad_check=gc) + _nin("hidden2_conv", "hidden2_conv_nin1", grad_check=gc)
    s += _pool("hidden2_conv_nin1", "hidden2_maxpool", 3, 2, 1) + _rnorm("hidden2_maxpool", "hidden2_rnorm")
    s += _conv("hidden2_rnorm", "hidden3_conv", 3, 1, 1, grad_check=gc) + _nin("hidden3_conv", "hidden3_conv_nin1", grad_check=gc)
    s += _conv("hidden3_conv_nin1", "hidden4_conv", 3, 1, 1, init_bias=1.0, grad_check=gc)
    s += _nin("hidden4_conv", "hidden4_conv_nin1", grad_check=gc) + _nin("hidden4_conv_nin1", "hidden4_conv_nin2", grad_check=gc)
    s += _conv("hidden4_conv_nin2", "hidden5_conv", 3, 1, 0, init_bias=1.0, grad_check=gc)
    s += _nin("hidden5_conv", "hidden5_conv_nin1", grad_check=gc) + _nin("hidden5_conv_nin1", "hidden5_conv_nin2", grad_check=gc)
    s += _pool("hidden5_conv_nin2", "hidden5_maxpool", 3, 2, 1)
    s += _fc("hidden5_maxpool", "hidden6", norm_limit=4, grad_check=gc) + _fc("hidden6", "hidden7", norm_limit=4, grad_check=gc)
    s += _fc("hidden7", "output", init_wt=0.1, norm_limit=4, grad_check=gc)
    return s


def mnist_conv(grad_check=False, image_size=28):
    gc = _gc(grad_check)
    s = _header("mnist_conv")
    s += _layer("input", 1, size=image_size)
    s += _layer("hidden1_conv", 48, "RECTIFIED_LINEAR") + _layer("hidden1_maxpool", 48)
    s += _layer("hidden2_conv", 128, "RECTIFIED_LINEAR") + _layer("hidden2_maxpool", 128)
    s += _layer("output", 10, "SOFTMAX")
    s += _conv("input", "hidden1_conv", 4, mom=0.95, tau=0, grad_check=gc) + _pool("hidden1_conv", "hidden1_maxpool", 4, 2)
    s += _conv("hidden1_maxpool", "hidden2_conv", 4, mom=0.95, tau=0, init_bias=1.0, grad_check=gc) + _pool("hidden2_conv", "hidden2_maxpool", 4, 2)
    s += _fc("hidden2_maxpool", "output", mom=0.95, tau=0, norm_limit=4, grad_check=gc)
    return s


def lenet5(grad_check=False):
    """LeNet-5-class net in the same schema (BASELINE configs[1]); average pooling exercises AvgPoolEdge."""
    gc = _gc(grad_check)
    s = _header("lenet5")
    s += _layer("input", 1, size=28)
    s += _layer("c1", 6, "RECTIFIED_LINEAR") + _layer("s2", 6) + _layer("c3", 16, "RECTIFIED_LINEAR") + _layer("s4", 16)
    s += _layer("f5", 120, "RECTIFIED_LINEAR") + _layer("f6", 84, "RECTIFIED_LINEAR") + _layer("output", 10, "SOFTMAX")
    s += _conv("input", "c1", 5, 1, 2, grad_check=gc) + _pool("c1", "s2", 2, 2, kind="AVERAGE_POOL")
    s += _conv("s2", "c3", 5, grad_check=gc) + _pool("c3", "s4", 2, 2, kind="MAXPOOL")
    s += _fc("s4", "f5", grad_check=gc) + _fc("f5", "f6", grad_check=gc) + _fc("f6", "output", grad_check=gc)
    return s


def vgg(image_size=224, num_classes=1000, widths=(64, 128, 256, 512, 512), depths=(2, 2, 3, 3, 3), dropprob=0.5):
    s = _header("vgg_style")
    s += _layer("input", 3, size=image_size)
    edges, prev = "", "input"
    for b, (w, d) in enumerate(zip(widths, depths), 1):
        for i in range(1, d + 1):
            name = f"conv{b}_{i}"
            s += _layer(name, w, "RECTIFIED_LINEAR")
            edges += _conv(prev, name, 3, 1, 1)
            prev = name
        s += _layer(f"pool{b}", w)
        edges += _pool(prev, f"pool{b}", 2, 2)
        prev = f"pool{b}"
    s += _layer("fc6", 4096, "RECTIFIED_LINEAR", dropprob) + _layer("fc7", 4096, "RECTIFIED_LINEAR", dropprob)
    s += _layer("output", num_classes, "SOFTMAX")
    edges += _fc(prev, "fc6") + _fc("fc6", "fc7") + _fc("fc7", "output")
    return s + edges


# forward MACs per image of a built net (for roofline accounting): see bench.py
def count_macs(net):
    """(fwd_macs, train_macs) per image following BASELINE.md §2: train = fwd + wgrad (all weighted
    edges) + dgrad (all but edges whose source is an input layer, src/convnet.cc:370)."""
    from .edge import ConvEdge, ConvOneToOneEdge, FCEdge
    fwd = train = 0
    for e in net.edges_:
        if isinstance(e, ConvEdge):
            d = e.conv_desc_
            macs = e.num_modules_y_ * e.num_modules_x_ * d.num_output_channels * d.kernel_size_y * d.kernel_size_x * d.num_input_channels
        elif isinstance(e, ConvOneToOneEdge):     # 1x1 conv: C x F per pixel
            macs = e.num_modules_y_ * e.num_modules_x_ * e.num_input_channels_ * e.num_output_channels_
        elif isinstance(e, FCEdge):
            macs = e._input_size() * e.num_output_channels_
        else:
            continue
        fwd += macs
        train += 2 * macs + (0 if e.GetSource().IsInput() or e.IsBackPropBlocked() else macs)
    return fwd, train
