"""Whole-net forward/backward on the CPU oracle, driven by a built ConvNet's graph and parameters
(sequential nets: every layer has one incoming edge).  Follows ConvNet::Fprop/Bprop order
(src/convnet.cc:377-405): per layer, ComputeOuter then ComputeDown of its outgoing edge, then
dropout' (none here) and activation'."""
import numpy as np

import oracle
from oracle import Geom


def _geom(e, src, N, pool=False):
    d = e.conv_desc_
    C, H, W = src.GetNumChannels(), src.GetSizeY(), src.GetSizeX()
    return Geom(N, C, H, W, C if pool else d.num_output_channels, d.kernel_size_y, d.kernel_size_x, d.stride_y, d.stride_x,
                -d.padding_y, -d.padding_x)


def forward_backward(net, x, labels, impl=None, force=None, dropout_states=None):
    """``dropout_states`` = {layer name: the device's post-dropout state}: the run is a TRAINING pass and a layer with
    dropprob > 0 applies the device's own Bernoulli mask (recovered as state != 0; the CPU cannot replay the GPU's RNG) with
    the train-time scale-up 1/(1-p) (src/layer.cc:367-397), and on the way back dropout' then ReLU' (layer.cc:399-413,556-558).
    ``force`` = (states, derivs) dicts of flat arrays taken from the device run: the BACKWARD ops are then each fed
    the device's own inputs (teacher forcing), so one ReLU unit or pool window that gates differently within fp32
    rounding cannot colour everything upstream of it — every op is still checked on realistic whole-net data."""
    from convnet_amd.edge import AvgPoolEdge, ConvEdge, ConvOneToOneEdge, FCEdge, MaxPoolEdge, ResponseNormEdge
    O = impl or oracle.port
    N = labels.size
    acts = {net.input_layers_[0].GetName(): np.ascontiguousarray(x.reshape(-1))}
    pre = {}
    for l in net.layers_:
        if l.IsInput():
            continue
        e = l.incoming_edge_[0]
        src = e.GetSource()
        a = acts[src.GetName()]
        if isinstance(e, ConvEdge):
            g = _geom(e, src, N)
            y = O.conv_up(g, a.reshape(g.in_shape()), e.GetWeight().ToNumpy().reshape(g.filt_shape()))
            y = O.add_row_vec(y.reshape(g.F, -1), e.GetBias().ToNumpy().reshape(-1)).reshape(-1)
        elif isinstance(e, MaxPoolEdge):
            g = _geom(e, src, N, True)
            y = O.max_pool(g, a.reshape(g.in_shape())).reshape(-1)
        elif isinstance(e, AvgPoolEdge):
            g = _geom(e, src, N, True)
            y = O.avg_pool(g, a.reshape(g.in_shape())).reshape(-1)
        elif isinstance(e, ResponseNormEdge):
            C = src.GetNumChannels()
            y = O.rnorm(a.reshape(C, -1, 1, N), e.num_filters_response_norm_, e.add_scale_, e.pow_scale_, e.blocked_).reshape(-1)
        elif isinstance(e, FCEdge):
            Fo = l.GetNumChannels()
            # CONV_ONETOONE = the same GEMM on the (N*X*Y, C) view: pixel and image together are the "case" axis
            Nv = a.size // src.GetNumChannels() if isinstance(e, ConvOneToOneEdge) else N
            y = O.dot(np.ascontiguousarray(a.reshape(-1, Nv)), e.GetWeight().ToNumpy(), np.zeros((Fo, Nv), np.float32), 0.0, 1.0, False, True)
            y = O.add_row_vec(y, e.GetBias().ToNumpy().reshape(-1)).reshape(-1)
        else:
            raise NotImplementedError(type(e))
        if l.is_relu:
            y = O.lower_bound(y, 0.0)
        if dropout_states is not None and l.dropprob_ > 0:
            assert l.is_relu and not l.store_dropout_noise_
            y = y * ((dropout_states[l.GetName()] != 0).astype(np.float32) * np.float32(1.0 / (1 - l.dropprob_)))
        if l.IsOutput():
            y = O.softmax_row_major(y.reshape(l.GetNumChannels(), N)).reshape(-1)
        acts[l.GetName()] = y
    out = net.output_layers_[0]
    f_acts, f_derivs = force if force is not None else (acts, None)
    derivs = {out.GetName(): O.softmax_grad_row_major(f_acts[out.GetName()].reshape(out.GetNumChannels(), N), labels).reshape(-1)}
    grads = {}
    for l in reversed(net.layers_):
        if l.IsOutput():
            continue
        e = l.outgoing_edge_[0]
        dst = e.GetDest()
        a, dy, yact = f_acts[l.GetName()], (f_derivs or derivs)[dst.GetName()], f_acts[dst.GetName()]
        dx = None
        if isinstance(e, ConvEdge):
            g = _geom(e, l, N)
            dw = O.conv_outp(g, a.reshape(g.in_shape()), dy.reshape(g.out_shape()), None, 0.0, e.scale_gradients_ / N)
            db = O.sum_by_axis(np.ascontiguousarray(dy.reshape(g.F, -1)), np.zeros(g.F, np.float32), 0, e.scale_gradients_ / N, 0.0)
            grads[e.GetName()] = (dw.reshape(-1), db)
            if not l.IsInput():
                dx = O.conv_down(g, dy.reshape(g.out_shape()), e.GetWeight().ToNumpy().reshape(g.filt_shape())).reshape(-1)
        elif isinstance(e, FCEdge):
            Nv = a.size // l.GetNumChannels() if isinstance(e, ConvOneToOneEdge) else N
            D, Fo = a.size // Nv, dst.GetNumChannels()
            a2, dy2 = np.ascontiguousarray(a.reshape(D, Nv)), np.ascontiguousarray(dy.reshape(Fo, Nv))
            dw = O.dot(dy2, a2, np.zeros((D, Fo), np.float32), 0.0, e.scale_gradients_ / N, True, False)
            db = O.sum_by_axis(dy2, np.zeros(Fo, np.float32), 0, e.scale_gradients_ / N, 0.0)
            grads[e.GetName()] = (dw.reshape(-1), db)
            if not l.IsInput():
                dx = O.dot(dy2, e.GetWeight().ToNumpy(), np.zeros((D, Nv), np.float32), 0.0, 1.0).reshape(-1)
        elif isinstance(e, MaxPoolEdge):
            g = _geom(e, l, N, True)
            dx = O.max_pool_undo(g, a.reshape(g.in_shape()), dy.reshape(g.pooled_shape()), yact.reshape(g.pooled_shape())).reshape(-1)
        elif isinstance(e, AvgPoolEdge):
            g = _geom(e, l, N, True)
            dx = O.avg_pool_undo(g, dy.reshape(g.pooled_shape())).reshape(-1)
        elif isinstance(e, ResponseNormEdge):
            C = l.GetNumChannels()
            dx = O.rnorm_undo(dy.reshape(C, -1, 1, N), a.reshape(C, -1, 1, N), e.num_filters_response_norm_, e.add_scale_, e.pow_scale_,
                              e.blocked_).reshape(-1)
        if dx is not None and not l.IsInput():
            if dropout_states is not None and l.dropprob_ > 0:
                dx = dx * np.float32(1.0 / (1 - l.dropprob_))
            if l.is_relu:
                dx = O.relu_deriv(dx, a)
            derivs[l.GetName()] = dx
    return acts, derivs, grads
